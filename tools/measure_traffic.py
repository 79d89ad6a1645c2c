"""HBM-side bytes per launch of the roofline-priced kernels, from rocprofv3 PMC passes, stamped with the hash
of the kernel sources so that bench.py only quotes numbers that belong to the build it runs.

    python tools/measure_traffic.py          (on the GPU box; writes profiles/pmc_traffic.json)

Method (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE and WRITE_SIZE in SEPARATE `--pmc` passes (TCC slot budget);
FETCH_SIZE counts 128-byte fabric requests at 64 bytes on gfx950 -> x2; both are reported in KiB by this
rocprofv3 -> x1024.  Per launch = counter summed over the kernel's dispatches / number of launches.
"""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DRIVER = r'''
import sys, torch
sys.path.insert(0, %r)
from eda_amd import attention
B, Lq, Lk = 8, 1024, 1024
q = torch.randn(B, Lq, 288, device="cuda", requires_grad=True); k = torch.randn(B, Lk, 288, device="cuda", requires_grad=True)
v = torch.randn(B, Lk, 288, device="cuda", requires_grad=True); w = torch.randn(B, Lq, 288, device="cuda")
for _ in range(4):
    o = attention.attention_core(q, k, v, None, 8, 0.1, 3); o.backward(w)
torch.cuda.synchronize()
''' % ROOT


DRIVER_SA1 = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from eda_amd import sa_ops, pointnet2_utils as PU
B, N, m, ns, C, chans, radius = 8, 50000, 2048, 64, 3, [64, 64, 128], 0.2
rng = np.random.default_rng(0)
xyz = torch.from_numpy(rng.uniform(-3, 3, (B, N, 3)).astype(np.float32)).cuda()
new_xyz = xyz[:, :m].contiguous()
idx = PU.ball_query(radius, ns, xyz, new_xyz)
chans = [3 + C] + chans
Ws = [torch.randn(chans[l + 1], chans[l], 1, 1, device="cuda").mul_(0.1).requires_grad_(True) for l in range(3)]
gs = [torch.ones(c, device="cuda", requires_grad=True) for c in chans[1:]]
bs = [torch.zeros(c, device="cuda", requires_grad=True) for c in chans[1:]]
running = [(torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")) for c in chans[1:]]
feats = torch.randn(B, N, C, device="cuda")
cfg = dict(gather=True, radius=radius, normalize_xyz=True, pool=ns, training=True, eps=1e-5, momentum=0.1, running=running)
params = []
for W, g, b in zip(Ws, gs, bs):
    params += [W, g, b]
for _ in range(4):
    out = sa_ops.FusedMLP.apply(cfg, None, xyz, new_xyz, feats, idx, *params)      # forward only
torch.cuda.synchronize()
''' % ROOT


DRIVER_SA1_EVAL = r'''
import sys, os
sys.path.insert(0, %r)
os.environ["ITERS"] = "4"
sys.argv = ["bench_sa_eval.py"]
sys.path.insert(0, os.path.join(%r, "tools"))
import bench_sa_eval
bench_sa_eval.main()
''' % (ROOT, ROOT)

DRIVER_GEMM = r'''
import sys, torch
sys.path.insert(0, %r)
from eda_amd import gemm
for R, K, N in ((2048, 288, 288), (8192, 288, 288)):
    x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    for _ in range(6):
        gemm.linear_fwd(x, w, b)
torch.cuda.synchronize()
''' % ROOT


def run_pass(counter, script):
    d = tempfile.mkdtemp(prefix="pmc_")
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, script],
                   check=True, capture_output=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, n = {}, {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
        n[k] = n.get(k, 0) + 1
    return {k: tot[k] / n[k] for k in tot}


def main():
    import bench
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(DRIVER)
        script = f.name
    fetch = run_pass("FETCH_SIZE", script)
    write = run_pass("WRITE_SIZE", script)
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(DRIVER_SA1)
        script_sa1 = f.name
    fetch_sa, write_sa = run_pass("FETCH_SIZE", script_sa1), run_pass("WRITE_SIZE", script_sa1)
    sa_kernels = ("gemm_gather3_kernel", "gemm_stream_kernel", "gemm_rows_kernel", "bn_relu_pool_kernel")
    sa_parts = {k: (2.0 * fetch_sa.get(k, 0.0) + write_sa.get(k, 0.0)) * 1024.0
                for k in set(fetch_sa) | set(write_sa) if any(t in k for t in sa_kernels)}

    def total(pred):
        return sum((2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0 for k in set(fetch) | set(write) if pred(k))
    entries = [
        {"op": "mha_fwd", "dims": [8, 8, 1024, 1024], "bytes_per_launch": total(lambda k: "mha_fwd_kernel" in k or "mha2_fwd_kernel" in k or "mha3_fwd_kernel" in k
                                                                                      or "mha4_fwd_kernel" in k)},
        {"op": "mha_bwd", "dims": [8, 8, 1024, 1024],
         "bytes_per_launch": total(lambda k: "mha_bwd_dq_kernel" in k or "mha_bwd_dkv_kernel" in k or "mha_part_reduce" in k
                                   or "mha2_bwd_kernel" in k or "mha2_part_reduce" in k)},
        # the fused SA1 forward = one launch per layer + pooling: sum of its kernels' per-launch bytes
        {"op": "sa_fused_fwd", "dims": [1048576, 64, 1, 6, 64, 64, 128], "bytes_per_launch": sum(sa_parts.values()),
         "kernels": {k[:90]: v for k, v in sorted(sa_parts.items())}},
    ]
    # round 5: the one-launch inference SA1 (csrc/sa_eval.hip) and the dominant critical-path row product
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(DRIVER_SA1_EVAL)
        script_ev = f.name
    fetch_ev, write_ev = run_pass("FETCH_SIZE", script_ev), run_pass("WRITE_SIZE", script_ev)
    ev = [k for k in set(fetch_ev) | set(write_ev) if "sa_eval_kernel" in k]
    if ev:
        entries.append({"op": "sa_fused_eval", "dims": [8 * 2048 * 64, 64, 6, 64, 64, 128],
                        "bytes_per_launch": sum((2.0 * fetch_ev.get(k, 0.0) + write_ev.get(k, 0.0)) * 1024.0 for k in ev)})
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(DRIVER_GEMM)
        script_g = f.name
    fetch_g, write_g = run_pass("FETCH_SIZE", script_g), run_pass("WRITE_SIZE", script_g)
    # (both shapes run the same 32 x 32-tile kernel template; the two launches of a process are told apart by grid size:
    # per-launch averages are taken per (kernel name) -- so one process per shape)
    for (R_, K_, N_) in ((2048, 288, 288),):
        drv = DRIVER_GEMM.replace("((2048, 288, 288), (8192, 288, 288))", "((%d, %d, %d),)" % (R_, K_, N_))
        with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
            f.write(drv)
            sg = f.name
        fg, wg = run_pass("FETCH_SIZE", sg), run_pass("WRITE_SIZE", sg)
        gk = [k for k in set(fg) | set(wg) if "gemm_rows_kernel" in k or "gemm_dma_kernel" in k]
        if gk:
            entries.append({"op": "gemm_fwd", "dims": [R_, K_, N_],
                            "bytes_per_launch": sum((2.0 * fg.get(k, 0.0) + wg.get(k, 0.0)) * 1024.0 for k in gk),
                            "kernels": {k[:90]: (2.0 * fg.get(k, 0.0) + wg.get(k, 0.0)) * 1024.0 for k in gk}})
    out = {"source_hash": bench.source_hash(),
           "how": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE in separate passes, KiB -> bytes, per launch "
                  "(tools/measure_traffic.py)", "entries": entries}
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)       # (gpurun only carries gpurun_out/ back)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
