#!/usr/bin/env python
"""Headline benchmark: scenes/s of one training step (forward + backward +
optimizer) of BeaUTyDETR on synthetic ScanRefer-shaped batches.

Workload (BASELINE.json metric "scenes/sec fwd+bwd (50k pts, 256 queries, 80 tok)"):
8 scenes per GPU, 50 000 points x (xyz+rgb), 256 queries, 80 text tokens, 132
detected boxes (butd on), fp32, random-init weights (no checkpoints offline),
frozen random-init RoBERTa-base; synthetic scalar loss touching every trainable
parameter.  Inputs are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One JSON line on rank 0 (contract in the task statement) carrying `roofline` and
`cpu_baseline` objects.  N > 1: scenes shard data-parallel (8 per rank, weak
scaling); gradients are summed with ONE RCCL all-reduce of a flat fp32 buffer.
"""
import os

# ROCm 7.2: replaying a captured HIP graph through the runtime's pre-recorded AQL packets ("graph
# packet capture") is not equivalent to launching its nodes once a host synchronisation has happened
# between two replays: the loss of this very step then CLIMBS (82 -> 110 in 12 un-synchronised replays
# where node-by-node launches give 82 -> 16; rocprofv3 shows 5-7x more back-to-back kernel pairs
# overlapping by 0.1-0.8 us).  Node-by-node launches cost nothing measurable (28.2 vs 28.1 ms/step), so
# the optimisation is switched off before the HIP runtime initialises.  DESIGN.md §4 has the evidence.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA (v_mfma_f32_16x16x4_f32) dense peak
MFMA_F32_MEASURED_TFLOPS = 126.0     # 2048 FLOP / 36 cycles x 1024 SIMDs x 2.17 GHz (profiles/r06_mha4_keys_per_wave.md)

SA_LEVELS = [  # (N, npoint, radius, nsample, C_feat) -- models/backbone_module.py:44-78
    (50000, 2048, 0.2, 64, 3), (2048, 1024, 0.4, 32, 128), (1024, 512, 0.8, 16, 256), (512, 256, 1.2, 16, 256)]


def algorithmic_bytes(name):
    """Algorithmic HBM bytes of one launch of a native op, keyed like
    eda_amd.ext._timed: (op, dims...).  Formulas of SURVEY.md §8d."""
    op, d = name[0], name[1:]
    if op == "ball_query":
        b, n, m, ns = d
        return b * (12 * n + 12 * m + 4 * m * ns)
    if op == "group_points":
        b, c, n, m, ns = d
        return b * (4 * c * n + 4 * m * ns + 4 * c * m * ns)
    if op == "group_points_grad":
        b, c, n, m, ns = d
        return b * (4 * c * n + 4 * m * ns + 4 * c * m * ns)
    if op == "furthest_point_sampling":
        b, n, m = d
        return b * (12 * n + 4 * m)
    if op in ("gather_points", "gather_points_grad"):
        b, c, n, m = d
        return b * (4 * c * m * 2 + 4 * m)
    if op == "three_nn":
        b, n, m = d
        return b * (12 * n + 12 * m + 24 * n)
    if op in ("group_concat_cl", "group_concat_cl_grad"):
        b, n, m, ns, c = d      # fused QueryAndGroup: xyz + features read once, rows written once
        return b * (12 * n + 12 * m + 4 * m * ns + 4 * c * n + 4 * (3 + c) * m * ns)
    if op == "bn_relu_fwd":
        r, c, pool, _ = d       # read z once, write the (pooled) activation once
        return 4 * r * c + 4 * (r // pool) * c
    if op == "bn_relu_bwd":
        r, c, pool, _ = d       # read z and d(out), write dz
        return 4 * r * c * 2 + 4 * (r // pool) * c
    if op == "sa_fused_fwd":
        # SURVEY 8d "fused SA layer fwd": inputs once + weights + idx + pooled output; the grouped tensor and the
        # activated tensors are never written.  (The pre-activations z_l kept for the backward are reported
        # separately as saved_bytes by fused_saved_bytes().)
        r, pool, _train, *ch = d
        w = sum(ch[i] * ch[i + 1] for i in range(len(ch) - 1))
        lvl = [v for v in SA_LEVELS if 3 + v[4] == ch[0] and 8 * v[1] * v[3] == r]
        if lvl and pool > 1:
            n, m, _, ns, c = lvl[0]
            return 8 * (12 * n + 4 * c * n + 12 * m + 4 * m * ns + 4 * ch[-1] * m) + 4 * w
        return 4 * r * ch[0] + 4 * w + 4 * (r // pool) * ch[-1]
    if op in ("three_interpolate", "three_interpolate_grad"):
        b, c, m, n = d if op == "three_interpolate" else (d[0], d[1], d[3], d[2])
        return b * (4 * c * m + 24 * n + 4 * c * n)
    return 0


def algorithmic_flops(name):
    """Useful QK^T + PV FLOPs of the fused attention ops (head_dim 36; padded columns not counted).
    Backward: dV, dP, dQ, dK plus the recomputed scores = 5 contractions of 2*Lq*Lk*36 per head."""
    op, d = name[0], name[1:]
    if op == "mha_fwd":
        b, h, lq, lk = d
        return 4.0 * b * h * lq * lk * 36
    if op == "mha_qproj_fwd":                 # q-projection (2 * rows * 288 * 288) + QK^T + PV in one launch
        b, h, lq, lk = d
        return 4.0 * b * h * lq * lk * 36 + 2.0 * b * lq * (h * 36) * (h * 36)
    if op == "mha_bwd":
        b, h, lq, lk = d
        return 10.0 * b * h * lq * lk * 36
    if op in ("gemm_fwd", "gemm_dgrad", "linear_add_dropout_ln_fwd"):
        r, k, n = d
        return 2.0 * r * k * n
    if op in ("gemm_grouped_fwd", "gemm_grouped_dgrad"):
        _g, r, k, n = d                     # (groups, rows, K, sum of the groups' widths)
        return 2.0 * r * k * n
    if op in ("sa_fused_fwd", "sa_fused_bwd"):
        r, _pool, _train, *ch = d
        f = 2.0 * r * sum(ch[i] * ch[i + 1] for i in range(len(ch) - 1))
        return f if op == "sa_fused_fwd" else 2.0 * f      # dX + dW (the first layer's dX only where it is needed)
    if op == "wgrad_grouped" and len(d) >= 4:
        return 2.0e6 * d[3]                  # (targets, jobs, tiles, 10^6 multiply-adds of all dW = dY^T X)
    if op == "gemm_frozen_b3":               # the frozen text encoder's wide layers (rows, K, N)
        r, k, n = d
        return 2.0 * r * k * n
    if op == "mha_fwd_hd64":                 # the frozen text encoder's attention, head_dim 64
        b, h, lq, lk = d
        return 4.0 * b * h * lq * lk * 64
    return 0.0


def fused_saved_bytes(name):
    """Bytes of pre-activations a fused SA / FP forward keeps for its backward (z_l of every layer)."""
    if name[0] != "sa_fused_fwd":
        return 0
    r, _pool, _train, *ch = name[1:]
    return 4 * r * sum(ch[1:])


def source_hash():
    """sha256 over the HIP sources: stamps measured-offline numbers (PMC traffic) to the build they belong to."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "eda_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "eda_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_pmc_traffic():
    """profiles/pmc_traffic.json (written by tools/measure_traffic.py from rocprofv3 PMC passes) -> {(op, dims):
    bytes per launch}; entries measured on a different build of the kernels are dropped (traffic: null)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return {}, "no profiles/pmc_traffic.json"
    if d.get("source_hash") != source_hash():
        return {}, "stale: profiles/pmc_traffic.json was measured on another build of csrc/ (%s)" % d.get("source_hash")
    return {(e["op"], tuple(e["dims"])): e["bytes_per_launch"] for e in d.get("entries", [])}, d.get("how", "")


def synthetic_loss(end_points):
    """Scalar that reaches every trainable parameter the real loss reaches, with
    no host sync (SURVEY.md §8d 'Loss for bwd')."""
    proj_tokens = end_points["proj_tokens"]
    prefixes = [k[:-len("center")] for k in end_points if k.endswith("center")]   # proposal_, {i}head_, last_
    # mean(seed logits^2) + sum over the prediction heads p of
    #   mean_bq |center_p|^2 + mean_bq |pred_size_p|^2 + mean(sem_cls_scores_p^2) + mean(proj_queries_p proj_tokens^T)
    loss = end_points["seeds_obj_cls_logits"].pow(2).mean() if os.environ.get("EDA_BENCH_LOSS_FORM") in ("perhead", "stacked") else None
    if os.environ.get("EDA_BENCH_LOSS_FORM") == "perhead":      # the original evaluation order (debugging aid)
        for p in prefixes:
            loss = loss + end_points[f"{p}center"].pow(2).sum(-1).mean() + end_points[f"{p}pred_size"].pow(2).sum(-1).mean()
            loss = loss + end_points[f"{p}sem_cls_scores"].pow(2).mean()
            loss = loss + torch.matmul(end_points[f"{p}proj_queries"], proj_tokens.transpose(1, 2)).mean()
        return loss

    def stacked(name):
        return torch.stack([end_points[p + name] for p in prefixes])
    if os.environ.get("EDA_BENCH_LOSS_FORM") == "stacked":      # the round 1-5 evaluation: one pow / sum / scale per term
        centers, sizes, sem, pq = stacked("center"), stacked("pred_size"), stacked("sem_cls_scores"), stacked("proj_queries")
        bq = float(centers.shape[1] * centers.shape[2])
        loss = loss + centers.pow(2).sum() / bq + sizes.pow(2).sum() / bq + sem.pow(2).sum() / (bq * sem.shape[-1])
        sim = torch.matmul(pq, proj_tokens.transpose(1, 2))              # (P, B, Q, L)
        return loss + sim.sum() / (bq * sim.shape[-1])
    # Round 6: the same sum of weighted squares as ONE weighted dot product over the concatenation of all squared terms
    # (seed logits / numel, centres and sizes / (B Q), class scores / (B Q C)): cat + mul + dot instead of ~18 launches of
    # 5 us each forward and as many backward (every term is `weight * sum(x^2)`; the weights are a constant vector)
    seeds = end_points["seeds_obj_cls_logits"]
    parts = [seeds] + [end_points[p + n] for n in ("center", "pred_size", "sem_cls_scores") for p in prefixes]
    c0 = end_points[prefixes[0] + "center"]
    bq = float(c0.shape[0] * c0.shape[1])
    cw = float(end_points[prefixes[0] + "sem_cls_scores"].shape[-1])
    wts = [1.0 / seeds.numel()] + [1.0 / bq] * (2 * len(prefixes)) + [1.0 / (bq * cw)] * len(prefixes)
    v = torch.cat([t.reshape(-1) for t in parts])
    w = _loss_weight_vector(tuple(t.numel() for t in parts), tuple(wts), v.device)
    pq = stacked("proj_queries")
    sim = torch.matmul(pq, proj_tokens.transpose(1, 2))                  # (P, B, Q, L)
    return torch.dot(v * w, v) + sim.sum() / (bq * sim.shape[-1])


_loss_w_cache = {}


def _loss_weight_vector(sizes, weights, device):
    """Constant per-element weights of synthetic_loss's squared terms (built once per layout, outside any capture)."""
    key = (sizes, weights, str(device))
    w = _loss_w_cache.get(key)
    if w is None:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("synthetic_loss: run one eager step before capturing (its weight vector is built on first use)")
        w = _loss_w_cache[key] = torch.cat([torch.full((n,), wt, dtype=torch.float32) for n, wt in zip(sizes, weights)]).to(device)
    return w


def make_targets(rank, per_gpu, device, inputs):
    """Synthetic ground truth for the real training loss (eda_amd/losses.py), resident on the device."""
    from eda_amd import synthetic
    t = synthetic.grounding_targets(rank, per_gpu, inputs["point_clouds"][..., :3].cpu().numpy(),
                                    inputs["tokenized"]["attention_mask"].cpu().numpy())
    return {k: torch.from_numpy(v).to(device) for k, v in t.items()}


def make_inputs(rank, per_gpu, device, n_points, max_len):
    from eda_amd import synthetic
    seeds = [rank * per_gpu + i for i in range(per_gpu)]
    pc = torch.from_numpy(synthetic.batch(seeds, n_points)).to(device)
    ids, am = synthetic.utterance_tokens(rank, per_gpu, max_len=max_len)
    boxes, bmask, cls = synthetic.detected_boxes(rank, per_gpu)
    return {
        "point_clouds": pc,
        "tokenized": {"input_ids": torch.from_numpy(ids).to(device),
                      "attention_mask": torch.from_numpy(am).to(device)},
        "det_boxes": torch.from_numpy(boxes).to(device),
        "det_bbox_label_mask": torch.from_numpy(bmask).to(device),
        "det_class_ids": torch.from_numpy(cls).to(device),
    }


def cpu_baseline(args):
    """The same training step on the host cores: this repo's model on torch CPU with
    the C oracle (oracle/, the CPU restatement of the reference's CUDA ops -- the
    reference has no CPU op path) as op backend.  Bounded sample: 1 step of 2 scenes."""
    from oracle import oracle_ext
    from eda_amd import pointnet2_utils
    from eda_amd.bdetr import BeaUTyDETR
    cores = min(os.cpu_count() or 1, args.cpu_threads)
    torch.set_num_threads(cores)
    oracle_ext.set_threads(cores)

    class MT:   # multi-threaded oracle entry points for the three SA-stack ops
        def __getattr__(self, k):
            return getattr(oracle_ext, k)

        def furthest_point_sampling(self, p, m):
            return oracle_ext.furthest_point_sampling(p, m, mt=True)

        def ball_query(self, a, b, r, ns):
            return oracle_ext.ball_query(a, b, r, ns, mt=True)

        def group_points(self, p, i):
            return oracle_ext.group_points(p, i, mt=True)

    from eda_amd import attention
    from oracle import attention_ref
    saved = pointnet2_utils._ext
    saved_core = attention._core
    pointnet2_utils._ext = MT()
    attention._core = attention_ref.attention_core
    try:
        scenes = args.cpu_scenes
        torch.manual_seed(0)
        model = BeaUTyDETR(num_queries=args.queries, butd=True).train()
        inputs = make_inputs(0, scenes, "cpu", args.points, args.tokens)
        t0 = time.time()
        loss = synthetic_loss(model(inputs))
        loss.backward()
        dt = time.time() - t0
    finally:
        pointnet2_utils._ext = saved
        attention._core = saved_core
        torch.set_num_threads(max(1, cores // 2))
    model_name = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model_name = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(scenes / dt, 4), "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": f"1 fwd+bwd step of {scenes} scenes ({args.points} pts, {args.queries} queries, "
                      f"{args.tokens} tok) on torch-CPU + oracle ops, {dt:.1f} s, no optimizer step",
            "cpu": model_name}


def run_cpu_baseline(args):
    """Host-CPU leg in a child process (own thread pools, hard time limit)."""
    import subprocess
    if args.cpu_scenes <= 0:
        return {"value": None, "unit": "scenes/s", "cores": None, "kind": "port", "sample": "skipped (--cpu-scenes 0)"}
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only",
           "--cpu-scenes", str(args.cpu_scenes), "--cpu-threads", str(args.cpu_threads),
           "--points", str(args.points), "--queries", str(args.queries), "--tokens", str(args.tokens)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_timeout, env=env)
        for line in r.stdout.splitlines():
            if line.startswith("CPU_BASELINE "):
                return json.loads(line[len("CPU_BASELINE "):])
        return {"value": None, "unit": "scenes/s", "cores": None, "kind": "port",
                "sample": "cpu leg failed: " + (r.stderr or "")[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "scenes/s", "cores": None, "kind": "port",
                "sample": f"cpu leg exceeded {args.cpu_timeout} s and was stopped"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--per-gpu", type=int, default=8, help="scenes per GPU")
    ap.add_argument("--points", type=int, default=50000)
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--tokens", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scenes", type=int, default=4)
    ap.add_argument("--cpu-threads", type=int, default=64, help="cap on host threads of the CPU leg")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="internal: run only the host-CPU leg and print its JSON object")
    ap.add_argument("--cpu-timeout", type=int, default=150)
    ap.add_argument("--no-butd", action="store_true")
    ap.add_argument("--blas", choices=["default", "rocblas", "hipblaslt"], default="default",
                    help="torch.backends.cuda.preferred_blas_library for the library GEMMs")
    ap.add_argument("--loss", choices=["synthetic", "hungarian"], default="synthetic",
                    help="synthetic: the sync-free scalar of SURVEY 8d (the headline metric); hungarian: the "
                         "reference's training loss with the matching solved on the device (eda_amd/losses.py)")
    ap.add_argument("--gemm-tuning", choices=["shipped", "online", "record", "off"], default="online",
                    help="EDA_FAST_ROBERTA=0 comparison line only -- library-GEMM selection through TunableOp "
                         "(tools/gemm_tuning.py): shipped = the results in tools/tuned only; online = those + tune unseen shapes during warm-up; "
                         "record = tune everything and write gpurun_out/tunableop_<ordinal>.csv; off = library default")
    ap.add_argument("--defer-wgrad", type=int, default=1,
                    help="1: queue the pointwise layers' weight gradients during the backward and compute them "
                         "in one grouped kernel (eda_amd/wgrad_queue.py); 0: compute each where autograd reaches it")
    ap.add_argument("--deterministic", type=int, default=0,
                    help="1: ordered per-owner sums instead of fp32 atomics in every gradient scatter "
                         "(eda_amd/deterministic.py; the reference's cudnn.deterministic, train_dist_mod.py:342-344)")
    ap.add_argument("--overlap", action="store_true",
                    help="run the text encoder on a side stream underneath the point backbone (measured slower)")
    ap.add_argument("--text-stream", type=int, default=1,
                    help="1 (with --graph): the frozen text encoder runs as its own HIP graph on a second stream "
                         "underneath the point backbone's graph (whose furthest point sampling keeps ~100 of the 256 "
                         "CUs busy for 3 ms); the rest of the step is a third graph behind an event.  0: one graph")
    ap.add_argument("--fps-prefetch", type=int, default=2, choices=[0, 1, 2],
                    help="1 (with --text-stream): the furthest point sampling of SA1 -- a function of the input coordinates "
                         "only, 3 ms of dependent rounds on ~100 CUs -- runs for the NEXT step's batch on the second stream "
                         "while the current step trains (an input pipeline has batch i+1 resident by then) and is handed to "
                         "the model through the reference's own `inds` argument; one sampling per timed step either way.  "
                         "2 (default): everything the backbone derives from the coordinates alone (the four samplings, four "
                         "ball queries, two 3-NN searches: Pointnet2Backbone.geometry; handed over through the backbone's "
                         "optional `geometry` argument) for the next batch on the second stream, once per timed step.  "
                         "0: all of it inside the step, on its critical path")
    ap.add_argument("--text-prefetch", type=int, default=1,
                    help="1 (with --fps-prefetch): the frozen text encoder runs for the NEXT step's tokens, behind the sampling on "
                         "the second stream (once per step either way); 0: for the current step, underneath the point backbone")
    ap.add_argument("--batches", type=int, default=2,
                    help="distinct synthetic batches resident in HBM, trained in turn (a loader would copy batch i+1 into "
                         "the static buffers while step i runs; here the copy is device-to-device): the pipelined step is "
                         "measured with a CHANGING batch.  1 = the same batch every step")
    ap.add_argument("--in-step-steps", type=int, default=8,
                    help="N = 1, default launch structure: additionally time this many steps of the un-pipelined structure "
                         "(one graph, sampling and text encoder inside the step: --text-stream 0 --fps-prefetch 0) and "
                         "report them under `in_step` of the same line; 0 = skip")
    ap.add_argument("--attn-dtype", choices=["f32", "bf16", "f16"], default="f32",
                    help="arithmetic of the attention QK^T / PV contractions: f32 = the headline / parity path; bf16 / "
                         "f16 = 16-bit MFMA with fp32 accumulation (csrc/mha2.hip with packed 16-bit quads, BASELINE.json configs[2] / [4]) -- "
                         "a SEPARATE bench line, never the headline")
    ap.add_argument("--sync-bn", nargs="?", const="native", default=None, choices=["native", "collective"],
                    help="N > 1: global-batch BatchNorm statistics like the reference's SyncBatchNorm (main_utils.py:336-338). "
                         "native (default when the flag is given): the BatchNorm kernels exchange their sums through "
                         "peer-mapped memory themselves (csrc/peer.h) -- no collective, every site stays fused, the step is "
                         "captured; collective: one packed all-reduce per BN layer and direction (eager launches, small-row "
                         "sites on torch ops).  Without the flag: per-GPU statistics (DESIGN.md §5)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="N > 1: nccl (= RCCL over xGMI, the measured configuration) or gloo on device tensors (debugging)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="N > 1 on a ONE-GPU box: every rank uses device 0 and the collectives go through gloo -- exercises "
                         "the multi-rank step structure end to end; the line it prints says so and is not a measurement")
    ap.add_argument("--overlap-allreduce", type=int, default=0,
                    help="N > 1: 1 = all-reduce the first half of the flat gradient buffer underneath the grouped "
                         "weight-gradient kernel of the second half (FlatParams.flush_and_reduce); 0 = one all-reduce "
                         "after the whole backward")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1: initialise a ONE-rank RCCL process group and run the step exactly as at N > 1 (bucket sampler "
                         "policy, split graphs, the two-range all-reduce through RCCL between them, optional --sync-bn "
                         "collectives inside the captured graphs): the multi-GPU code path on the one GPU a test box has")
    ap.add_argument("--split-graphs", action="store_true",
                    help="use the N>1 graph structure (two graphs, eager all-reduce slot) even at N=1")
    ap.add_argument("--kernel-steps", type=int, default=3,
                    help="eager steps (after the timed region) used for the per-kernel event timings in graph mode")
    ap.add_argument("--text-encoder-mode", choices=["train", "eval"], default="train",
                    help="train (default, the reference: main_utils.py:459 puts the WHOLE model in train mode, so the frozen RoBERTa "
                         "runs with its dropout layers active) or eval (no dropout inside the frozen encoder: what rounds 1-5 timed)")
    ap.add_argument("--graph", type=int, default=1,
                    help="1 = capture the whole training step in a HIP graph and replay it (default); "
                         "0 = eager launches")
    args = ap.parse_args()

    if args.cpu_baseline_only:
        print("CPU_BASELINE " + json.dumps(cpu_baseline(args)))
        return

    t_start = time.perf_counter()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, rendezvous on 127.0.0.1)
        import socket
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("[bench] --gpus %d without WORLD_SIZE: re-launching as %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path is HIP-only (no CPU fallback)")
    if args.share_gpu:
        local_rank = 0                  # functional test of the N > 1 structure on a one-GPU box (never a measurement)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist_on = world > 1 or args.force_dist       # the step has collectives in it
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1])); s_.close()
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.share_gpu or args.dist_backend == "gloo":
            dist.init_process_group("gloo")         # (RCCL refuses two ranks on one device)
        else:
            dist.init_process_group("nccl", device_id=device)
        from eda_amd import parallel as _par
        _par.reserve_cus_for_collectives(32)        # the prefetched sampling runs next to RCCL's channel workgroups
        if os.environ.get("EDA_FPS_BUCKET") is None:
            # ... and on the sampler whose workgroups never wait for each other (5 ms on 8 CUs of the second stream instead
            # of 3 ms on 104: hidden under the step either way, 20.73 vs 20.64 ms/step at N = 1)
            _par.sampler_without_co_residency()

    from eda_amd import ext
    if args.sync_bn == "collective" and dist_on and args.graph:
        # SyncBN puts collectives INSIDE the step; a captured collective's events are queried by RCCL's watchdog thread on this
        # torch / ROCm ("operation not permitted on an event last recorded in a capturing stream": the process aborts), so this
        # configuration runs with eager launches.  The default N > 1 step (per-GPU statistics) keeps its graphs: its one
        # all-reduce is launched between them.
        print("[bench] --sync-bn with a process group: eager launches (collectives inside a captured step abort RCCL's watchdog)",
              file=sys.stderr, flush=True)
        args.graph = 0
    if args.sync_bn and dist_on:
        from eda_amd import sync_bn
        if args.sync_bn == "collective" and dist.get_backend() == "gloo" and args.graph:
            log_early = lambda m: print("[bench] " + m, file=sys.stderr, flush=True)
            log_early("gloo collectives cannot be captured in a HIP graph: --sync-bn collective with gloo runs eager (--graph 0)")
            args.graph = 0
        sync_bn.enable(single_rank_too=(world == 1), native=(args.sync_bn == "native"))        # fused SA / FP calls keep their fusion (library hook); no host synchronisation anywhere, so the
        # step is captured like the default one (a capture failure falls back to eager launches below)
        if args.sync_bn == "native" and not sync_bn.native() and args.graph:
            # the peer-memory self-test failed on some rank (sync_bn logged it): the statistics go through collectives, which
            # cannot live inside a captured step (above)
            print("[bench] --sync-bn native fell back to collectives: eager launches", file=sys.stderr, flush=True)
            args.graph = 0
    from eda_amd.bdetr import BeaUTyDETR
    from eda_amd.parallel import FlatParams, reference_lr_groups
    if args.blas != "default":
        torch.backends.cuda.preferred_blas_library("cublas" if args.blas == "rocblas" else "cublaslt")
    if args.deterministic:
        from eda_amd import deterministic
        deterministic.enable(True)
    tuning_file, shipped_ok = None, False
    fast_roberta = os.environ.get("EDA_FAST_ROBERTA", "1") != "0"
    if fast_roberta:
        args.gemm_tuning = "off"        # no library GEMM is left in the step (eda_amd/roberta_fast.py): nothing to select
    if args.gemm_tuning != "off":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import gemm_tuning
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        tuning_file, shipped_ok = gemm_tuning.enable(
            online=args.gemm_tuning in ("online", "record"),
            scratch=os.path.join(ROOT, "gpurun_out", "tunableop_.csv"),
            use_shipped=args.gemm_tuning != "record")

    torch.manual_seed(0)                       # same init on every rank (DDP broadcast equivalent)
    model = BeaUTyDETR(num_queries=args.queries, butd=not args.no_butd).to(device).train()
    if args.text_encoder_mode == "eval":       # (the reference: model.train() on everything, main_utils.py:459 -- the frozen
        model.text_encoder.eval()              #  encoder's dropout layers are active while it trains; "eval": rounds 1-5's line)
    model.overlap_text_encoder = args.overlap
    no_drop = os.environ.get("EDA_BENCH_NO_DROPOUT", "")          # debugging aid: "1" (all), "modules", "attention"
    if no_drop:
        for name_, mod_ in model.named_modules():
            if isinstance(mod_, torch.nn.Dropout) and (no_drop in ("1", "modules") or (no_drop == "ffn") == ("ffn" in name_)
                                                        and no_drop in ("ffn", "notffn")):
                mod_.p = 0.0
            if hasattr(mod_, "dropout") and isinstance(getattr(mod_, "dropout"), float) and no_drop in ("1", "attention"):
                mod_.dropout = 0.0
    flat = FlatParams(model, reference_lr_groups)
    lrs = {"base": 1e-4, "backbone_net": 1e-3, "text_encoder": 1e-5}     # scripts/train_scanrefer.sh
    opt = torch.optim.AdamW([{"params": [gp], "lr": lrs[k]} for k, gp in flat.groups.items()],
                            weight_decay=5e-4, fused=True, capturable=bool(args.graph))
    from eda_amd import attention, pipeline
    attention.set_compute_dtype(args.attn_dtype)

    # the synthetic batches of this rank (resident in HBM; for the real loss their targets ride along in the dict)
    nb = max(1, args.batches)
    batches, target_keys = [], []
    for k_ in range(nb):
        bt = make_inputs(rank + k_ * world, args.per_gpu, device, args.points, args.tokens)
        if args.loss == "hungarian":
            tg = make_targets(rank + k_ * world, args.per_gpu, device, bt)
            if dist_on:
                # the loss's normaliser is the GLOBAL batch's box count (losses.py:630-636): formed here, when the batch arrives,
                # so that the captured step holds no collective (a loader would do the same one batch ahead)
                from eda_amd import losses as _L
                tg["num_boxes_global"] = _L.global_box_count(tg["box_label_mask"])
            target_keys = sorted(tg)
            bt.update(tg)
        batches.append(bt)
    inputs = pipeline._clone(batches[0])       # static buffers of the un-pipelined step structures
    inputs_flat = pipeline._flat(inputs)
    batches_flat = [pipeline._flat(bt) for bt in batches]
    counter = [0]

    def load_batch():
        """What a loader does for the un-pipelined structures: batch i into the static input buffers before step i."""
        if nb > 1:
            torch._foreach_copy_(inputs_flat, batches_flat[counter[0] % nb])
        counter[0] += 1

    if args.loss == "hungarian":
        from eda_amd import losses as L
        parts = os.environ.get("EDA_LOSS_PARTS", "boxes,labels,contrastive_align,qp").split(",")   # (debug switch)
        criterion = L.SetCriterion(L.HungarianMatcher(1, 0, 2, True),
                                   losses=[x for x in ["boxes", "labels", "contrastive_align"] if x in parts],
                                   eos_coef=0.1, temperature=0.07)            # main_utils.py:264-271

        def loss_fn(end_points, batch):
            end_points.update({k: batch[k] for k in target_keys})
            end_points["language_dataset"] = ["scanrefer"] * args.per_gpu
            if "qp" not in parts:
                end_points.pop("seeds_obj_cls_logits")
            return L.compute_hungarian_loss(end_points, 6, criterion, query_points_obj_topk=4)[0]
    else:
        def loss_fn(end_points, batch):
            return synthetic_loss(end_points)

    ingraph_hist = None
    if os.environ.get("EDA_BENCH_INGRAPH_HIST") == "1":
        ingraph_hist = (torch.full((4096,), 0.0, device=device), torch.full((1,), 0, dtype=torch.long, device=device))

    # N > 1: the flat gradient is reduced in two ranges, the first one's all-reduce in flight underneath the second
    # range's grouped weight-gradient kernel (DDP overlaps its buckets with the backward, main_utils.py:343-346; here
    # everything the collective needs is produced at the very end of the backward)
    overlap_ar = dist_on and bool(args.overlap_allreduce) and bool(args.defer_wgrad)
    ar_mid, ar_total, ar_works = flat.split_offset(0.5), flat.flat_grad.numel(), []

    def backward(loss):
        if overlap_ar:
            with flat.deferred_wgrad(flush=False):
                loss.backward()
            flat.flush_range(0, ar_mid)              # range A complete: its collective can start
            return
        if args.defer_wgrad:
            with flat.deferred_wgrad():              # weight gradients: one grouped kernel after the backward
                loss.backward()
        else:
            loss.backward()
        flat.collect_grads()

    def reduce_a():
        ar_works.append(dist.all_reduce(flat.flat_grad[:ar_mid], async_op=True))

    def stage_b():
        flat.flush_range(ar_mid, ar_total)
        flat.finish_ranges()

    def reduce_b():
        ar_works.append(dist.all_reduce(flat.flat_grad[ar_mid:], async_op=True))
        for w_ in ar_works:
            w_.wait()
        ar_works.clear()                             # (the 1/world rides in the clip's multiplication, update())

    # Collectives issued BEFORE the step is captured run on a stream of their own that never captures.  RCCL's watchdog thread
    # polls the completion events of the collectives issued so far; synchronous collectives record those events on the stream
    # they were issued on, and on this ROCm an event query on a stream that has since entered capture fails with "operation
    # not permitted on an event last recorded in a capturing stream", which aborts the process from the watchdog thread (seen
    # in 2 of 6 --force-dist runs of round 4, worked around with a sleep until round 5).  Stream order is kept with
    # wait_stream on both sides; after the capture the collectives run between the graphs on the replay stream itself.
    coll_stream = torch.cuda.Stream() if dist_on else None
    pre_capture = [bool(args.graph)]

    def on_coll_stream(fn):
        if not (dist_on and pre_capture[0]):
            return fn()
        cur = torch.cuda.current_stream()
        coll_stream.wait_stream(cur)
        with torch.cuda.stream(coll_stream):
            fn()
        cur.wait_stream(coll_stream)

    def all_reduce_now():
        if overlap_ar:
            reduce_a(); stage_b(); reduce_b()
        elif dist_on:
            dist.all_reduce(flat.flat_grad)

    def all_reduce():
        on_coll_stream(all_reduce_now)

    def record_loss(loss):
        if ingraph_hist is not None:                 # debugging aid: loss history written by the graph itself
            ingraph_hist[0].index_copy_(0, ingraph_hist[1], loss.detach().reshape(1))
            ingraph_hist[1].add_(1)

    def fwd_bwd():
        attention.advance_dropout_state(device)      # new attention-dropout masks every step
        loss = loss_fn(model(inputs), inputs)
        backward(loss)
        record_loss(loss)
        return loss

    # clip + AdamW as two launches on the flat buffers (csrc/optim.hip; torch's optimizer object keeps the state);
    # EDA_FLAT_ADAMW=0: FlatParams.clip_grad_norm_ + torch's fused AdamW (~12 launches; rounds 1-6)
    flat_opt = None
    if os.environ.get("EDA_FLAT_ADAMW", "1") != "0":
        from eda_amd.parallel import FlatClipAdamW
        flat_opt = FlatClipAdamW(flat, opt)

    def update():
        if flat_opt is not None:
            flat_opt.step(0.1, 1.0 / world)                   # main_utils.py:483-486 (on the mean over ranks) + :277-305
            return
        flat.clip_grad_norm_(0.1, pre_scale=1.0 / world)
        opt.step()

    def core_step():
        loss = fwd_bwd()
        all_reduce()
        update()
        return loss

    def step():
        load_batch()
        return core_step()

    def sync_bn_native_on():
        from eda_amd import sync_bn as _s
        return _s.native()

    def _lib_peer_kind():
        from eda_amd import _lib as _l
        return _l.lib().eda_peer_alloc_kind()

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    log("model + inputs ready")
    eager_step = step
    if args.graph:
        # HIP graphs of the step: ~4000 launches are replayed instead of issued by the host.
        # Inputs live in static HBM buffers (a data loader would copy the next batch into
        # them).  N = 1: one graph holds forward, backward, clip and AdamW.  N > 1: the RCCL
        # all-reduce of the flat gradient buffer is launched eagerly between two graphs
        # (forward+backward | clip+AdamW) rather than captured.
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                eager_step()
        torch.cuda.current_stream().wait_stream(side)
        barrier_sync = (lambda: (on_coll_stream(dist.barrier) if dist_on else None, torch.cuda.synchronize()))
        barrier_sync()
        # (no collective's completion event sits on a stream that captures below: on_coll_stream above)
        # thread-local capture mode: calls from other threads (RCCL's watchdog) must not
        # invalidate the capture.
        mode = dict(capture_error_mode="thread_local")
        # (capture on the stream the eager warm-up steps ran on)
        try:
            if args.text_stream and not args.overlap:
                # three graphs on two streams (eda_amd/pipeline.py): [next batch's SA1 sampling + frozen text encoder] on a
                # second stream underneath [point backbone | rest of the forward, loss, backward (, clip + AdamW at N = 1)]
                pipe = pipeline.PipelinedTrainStep(
                    model, batches[0], loss_fn, backward, update, stream=side,
                    all_reduce=all_reduce if (dist_on and not overlap_ar) else None,
                    post_stages=[(None, reduce_a), (stage_b, reduce_b)] if overlap_ar else None,
                    split_update=(dist_on or args.split_graphs),
                    prefetch={0: None, 1: "sa1", 2: "geometry"}[args.fps_prefetch],
                    text_prefetch=bool(args.text_prefetch), after_loss=record_loss)
                static_loss = pipe.loss
                pcount = [0]

                def step():
                    pcount[0] += 1
                    return pipe.step(next_batch=batches[pcount[0] % nb] if nb > 1 else None)

            elif not dist_on and not args.split_graphs:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side, **mode):
                    static_loss = core_step()

                def step():
                    load_batch()
                    graph.replay()
                    return static_loss
            else:
                g_fb, g_up = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                pool = torch.cuda.graph_pool_handle()
                with torch.cuda.graph(g_fb, pool=pool, stream=side, **mode):
                    static_loss = fwd_bwd()
                g_sb = None
                if overlap_ar:
                    # range B's flush is queue state + launches that only exist while CAPTURING (the weight-gradient
                    # queue is filled by the captured backward): it has to be a graph of its own between the two
                    # collectives, as PipelinedTrainStep's post_stages are -- run eagerly after g_fb.replay() it would
                    # flush an empty queue from the second step on and leave range B at the captured zero fill
                    g_sb = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_sb, pool=pool, stream=side, **mode):
                        stage_b()
                with torch.cuda.graph(g_up, pool=pool, stream=side, **mode):
                    update()

                def step():
                    load_batch()
                    g_fb.replay()
                    if g_sb is not None:
                        reduce_a(); g_sb.replay(); reduce_b()
                    else:
                        all_reduce()
                    g_up.replay()
                    return static_loss
            pre_capture[0] = False
            log("step captured in HIP graph(s)")
        except Exception as exc:     # never lose the run to a capture problem: fall back to eager launches
            log(f"graph capture failed ({type(exc).__name__}: {exc}); falling back to eager launches")
            torch.cuda.synchronize()
            args.graph = 0
            pre_capture[0] = False
            step = eager_step

    trace_loss = os.environ.get("EDA_BENCH_TRACE_LOSS") == "1"      # debugging aid: loss of every warm-up step
    for i in range(args.warmup):
        l_ = step()
        if trace_loss:
            log("warmup step %d loss %.3f" % (i, float(l_.detach())))
    torch.cuda.synchronize()
    log("warmup done")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    timer = ext.OpTimer()
    barrier()
    ext.op_timer = timer
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        loss = step()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    ext.op_timer = None
    final_loss = float(loss.detach())          # loss of the last timed step (read before the eager kernel-timing runs below)
    fps_in_step_ms = [round(v, 3) for v in ext.fps_last_duration_ms(device)]
    log("furthest point sampling inside the timed steps, device-clock wall time of the last launch per workspace (ms): %s"
        % fps_in_step_ms)
    if ingraph_hist is not None:
        n_ = int(ingraph_hist[1].item())
        log("in-graph loss history (%d steps incl. eager warm-up): " % n_ + " ".join("%.1f" % v for v in ingraph_hist[0][:n_].tolist()))
    if os.environ.get("EDA_BENCH_TRACE_LOSS") == "1":
        with torch.no_grad():
            log("loss of the last timed step %.3f; eager forward with the trained parameters: %.3f"
                % (final_loss, float(loss_fn(model(inputs), inputs))))
    if args.graph:
        # graph replay runs no host code, so per-kernel HIP events cannot be interleaved with
        # it: time the native kernels in a few eager runs of the SAME step right after.
        for _ in range(2):
            eager_step()
        torch.cuda.synchronize()
        ext.op_timer = timer
        for _ in range(args.kernel_steps):
            eager_step()
        torch.cuda.synchronize()
        ext.op_timer = None
    # The same work with NOTHING moved out of the step: one graph, SA1's sampling and the text encoder of the batch being
    # trained inside it (the reference's loop structure) -- timed in the same run so that both schedules are on the
    # driver's clock
    in_step = None
    if args.graph and not dist_on and args.text_stream and not args.overlap and not args.split_graphs and args.in_step_steps > 0:
        try:
            g_one = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g_one, stream=side, capture_error_mode="thread_local"):
                core_step()
            for _ in range(3):
                load_batch()
                g_one.replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.in_step_steps):
                load_batch()
                g_one.replay()
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            in_step = {"value": round(args.per_gpu * args.in_step_steps / dt1, 3), "unit": "scenes/s",
                       "ms_per_step": round(dt1 / args.in_step_steps * 1e3, 3), "steps": args.in_step_steps,
                       "launch": "one hipGraph, one stream: furthest point sampling and text encoder of the batch being "
                                 "trained inside the step (= --text-stream 0 --fps-prefetch 0)"}
            log(f"in-step structure: {in_step['ms_per_step']} ms/step")
            del g_one
        except Exception as exc:
            log(f"in-step measurement failed ({type(exc).__name__}: {exc})")
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()

    fps_repaired = ext.fps_repaired(device)
    if fps_repaired:
        log(f"furthest point sampling: {fps_repaired} cluster launch(es) were not co-resident and were repaired on the device "
            "by the bucket sampler (indices exact; each cost its spin limit)")
    fps_giveups = ext.fps_status(device)
    if fps_giveups:
        raise SystemExit(f"furthest point sampling gave up its inter-workgroup spin in {fps_giveups} workspace(s): "
                         "the sampled indices of this run are not the FPS result (include/eda_hip.h)")
    if args.sync_bn and dist_on:
        # the in-kernel BatchNorm statistics exchange never hangs -- a poll that ran into its bound is COUNTED and the statistics
        # are garbage from then on (csrc/peer.h): a run with a non-zero count measured a broken training step
        from eda_amd import sync_bn as _sbn
        _sbn.check()
    log(f"timed region done: {dt / args.steps * 1e3:.1f} ms/step")
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        scenes = args.per_gpu * world * args.steps
        # launches tagged "side" (eda_amd.ext.tagged: the frozen text encoder) run on the second stream under the pipelined
        # schedule, i.e. off the critical path
        pipelined_sched = bool(args.graph and args.text_stream and not args.overlap)
        # ---- per-kernel numbers measured live with HIP events on the launch stream ----
        summ = timer.summary()
        ksteps = args.kernel_steps if args.graph else args.steps
        kernels = []
        for name, (calls, ms) in summ.items():
            byts = algorithmic_bytes(name)
            flops = algorithmic_flops(name)
            # a fused SA / FP forward in TRAINING mode must keep the pre-activations z_l (the BatchNorm backward
            # needs them densely): HBM bytes of the formulation = SURVEY 8d's fused figure + each z_l written
            # once and read once (by the next layer's operand staging / the pooling pass)
            byts_design = byts + 2 * fused_saved_bytes(name)
            kernels.append({"op": name[0], "dims": list(name[1:]), "stream": timer.tags.get(name, "main") if pipelined_sched else "main",
                            "calls_per_step": calls / ksteps,
                            "ms": round(ms, 4), "alg_bytes": byts, "saved_bytes": fused_saved_bytes(name),
                            "design_bytes": byts_design,
                            "gbs": round(byts_design / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                            "tflops": round(flops / (ms * 1e-3) / 1e12, 2) if flops and ms > 0 else None})
        kernels.sort(key=lambda k: -k["ms"] * k["calls_per_step"])
        native_ms = sum(k["ms"] * k["calls_per_step"] for k in kernels)
        # matrix work of ONE step (algorithmic FLOPs of every product-shaped native call, both streams) against the step's time:
        # how much of the fp32 matrix pipe the whole step keeps busy, whatever the launch structure
        mat = {"main": 0.0, "side": 0.0}
        for name, (calls, _ms) in summ.items():
            tag = timer.tags.get(name, "main") if pipelined_sched else "main"
            mat["side" if tag == "side" else "main"] += algorithmic_flops(name) * calls / ksteps
        # HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) of
        # THIS build of the kernels (tools/measure_traffic.py stamps them with the source hash); null otherwise
        pmc_traffic, traffic_how = load_pmc_traffic()
        # dominant HBM-priced kernel (FPS is latency-bound: reported as us/round below)
        # HBM-priced candidates: launches that move at least 32 MB (smaller ones are launch- or
        # latency-bound and are listed in `kernels` only); FPS is latency-bound and reported below.
        hbm = [k for k in kernels if k["design_bytes"] >= (32 << 20) and k["op"] != "furthest_point_sampling"]
        dom = hbm[0] if hbm else None
        # achievable HBM rate on this box: the library's copy kernel over 2 x 1 GiB (4x the 256 MB
        # Infinity Cache), read + write bytes / HIP-event time (report both denominators)
        copy_gbs = None
        try:
            from eda_amd import ext as _ext
            n_copy = 1 << 28
            src_c = torch.full((n_copy,), 1.0, device=device)
            dst_c = torch.empty_like(src_c)
            for _ in range(2):
                _ext.device_copy(src_c, dst_c)
            ce0, ce1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ce0.record()
            for _ in range(5):
                _ext.device_copy(src_c, dst_c)
            ce1.record()
            torch.cuda.synchronize()
            copy_gbs = round(2.0 * 4 * n_copy * 5 / (ce0.elapsed_time(ce1) * 1e-3) / 1e9, 1)
            del src_c, dst_c
        except torch.OutOfMemoryError:
            copy_gbs = None
        # ---- the same SA1 level in INFERENCE mode: one launch, nothing but the pooled output written (csrc/sa_eval.hip) --
        # SURVEY 8d's "fused SA layer fwd" bytes are reachable only there (training keeps every pre-activation for the
        # BatchNorm backward); priced on those bytes AND as matrix work (it is compute-bound: 1 166 FLOP per byte)
        roofline_hbm_eval = None
        try:
            from eda_amd import pointnet2_utils as _PU, sa_ops as _so
            sa1 = model.backbone_net.sa1
            with torch.no_grad():
                pcs = inputs["point_clouds"]
                xyz_e = pcs[..., :3].contiguous(); feats_e = pcs[..., 3:6].contiguous()
                inds_e = _PU.furthest_point_sample(xyz_e, sa1.npoint)
                nx_e = _PU.gather_operation(xyz_e.transpose(1, 2).contiguous(), inds_e).transpose(1, 2).contiguous()
                idx_e = _PU.ball_query(sa1.radius, sa1.nsample, xyz_e, nx_e)
                lay_e = sa1.mlp_module.layers(); bns_e = [l_.bn.bn for l_ in lay_e]
                cfg_e = dict(gather=True, radius=sa1.radius, normalize_xyz=True, pool=sa1.nsample, training=False, eps=bns_e[0].eps,
                             momentum=bns_e[0].momentum, running=[(b_.running_mean, b_.running_var) for b_ in bns_e])
                fn_e = lambda: _so._one_pass_eval(cfg_e, xyz_e, nx_e, feats_e, idx_e, lay_e, bns_e, sa1.nsample)
                if fn_e() is not None:
                    torch.cuda.synchronize()
                    ee0, ee1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ee0.record()
                    for _ in range(10):
                        fn_e()
                    ee1.record()
                    torch.cuda.synchronize()
                    ms_e = ee0.elapsed_time(ee1) / 10
                    Bq, mq, nsq = xyz_e.shape[0], sa1.npoint, sa1.nsample
                    alg_e = Bq * (12 * xyz_e.shape[1] + 12 * xyz_e.shape[1] + 12 * mq + 4 * (6 * 64 + 64 * 64 + 64 * 128) + 4 * mq * nsq + 4 * 128 * mq)
                    fl_e = 2.0 * Bq * mq * nsq * (6 * 64 + 64 * 64 + 64 * 128)
                    key_e = ("sa_fused_eval", (Bq * mq * nsq, nsq, 6, 64, 64, 128))
                    roofline_hbm_eval = {"kernel": "sa_fused_eval(%d rows, 6 -> 64 -> 64 -> 128, pool %d): one launch, inference" % (Bq * mq * nsq, nsq),
                                         "bound": "hbm (as SURVEY 8d prices it) / mfma (what it is: %.0f FLOP per byte)" % (fl_e / alg_e),
                                         "ms_per_launch": round(ms_e, 4), "alg_bytes_per_launch": alg_e,
                                         "achieved": round(alg_e / (ms_e * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": round(alg_e / (ms_e * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                         "traffic": pmc_traffic.get(key_e),
                                         "traffic_over_alg_bytes": round(pmc_traffic[key_e] / alg_e, 2) if pmc_traffic.get(key_e) else None,
                                         "fp32_equivalent_tflops": round(fl_e / (ms_e * 1e-3) / 1e12, 1),
                                         "frac_of_fp32_mfma_peak": round(fl_e / (ms_e * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 3),
                                         "arithmetic": "layer 1 fp32 MFMA; layers 2, 3 bf16 x 3 on v_mfma_f32_16x16x32_bf16 (fp32 accuracy)",
                                         "timing": "HIP events around 10 launches"}
        except Exception as exc:      # a measurement aid: never lose the line to it
            roofline_hbm_eval = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        roofline_hbm = None
        if dom:
            roofline_hbm = {"kernel": f"{dom['op']}{tuple(dom['dims'])}", "bound": "hbm",
                            "achieved": dom["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(dom["gbs"] / HBM_PEAK_GBS, 5),
                            # SURVEY 8d's own figure: the fused layer's ALGORITHMIC bytes (grouped tensor and
                            # activations never written) / time / peak -- by construction tiny in training, where every
                            # pre-activation is kept for the BatchNorm backward (see `note`)
                            "frac_algorithmic": round(dom["alg_bytes"] / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                            "achievable_copy_gbs": copy_gbs,
                            "guide_copy_gbs": 6290.0,       # MI355X_MICROARCH.md: measured float4 copy, 79 % of the 8 TB/s spec
                            "frac_of_achievable": round(dom["gbs"] / copy_gbs, 5) if copy_gbs else None,
                            "traffic": pmc_traffic.get((dom["op"], tuple(dom["dims"]))),
                            "traffic_over_alg_bytes": (round(pmc_traffic[(dom["op"], tuple(dom["dims"]))] / dom["alg_bytes"], 1)
                                                       if pmc_traffic.get((dom["op"], tuple(dom["dims"]))) and dom["alg_bytes"] else None),
                            "traffic_source": traffic_how,
                            "ms_per_launch": dom["ms"], "alg_bytes_per_launch": dom["alg_bytes"],
                            "saved_activation_bytes": dom["saved_bytes"], "priced_bytes_per_launch": dom["design_bytes"],
                            "note": "achieved = (SURVEY 8d fused bytes + pre-activations kept for the backward, written "
                                    "once and read once) / time of the whole fused call (one launch per layer + pooling)",
                            "ms_per_step": round(dom["ms"] * dom["calls_per_step"], 4)}
        # MFMA-priced single-kernel launches: attention and the own row GEMMs (multi-kernel calls such as
        # sa_fused_* and the grouped weight gradient are priced in `kernels`)
        # ... that run >= 50 us per launch: shorter launches (the 2048-row GEMMs of the decoder, 11 us each in the rocprofv3
        # trace) are bound by launch latency and the ~12 B/clk a CU can ingest, a HIP-event bracket around one of them
        # measures mostly the bracket, and no roof prices them; their family total is `gemm_family_ms_per_step` and
        # `launch_bound_gemm` below
        GEMM_OPS = ("gemm_fwd", "gemm_dgrad", "linear_add_dropout_ln_fwd")   # (the last: product + LayerNorm epilogue)
        mf_all = [k for k in kernels if k["tflops"] and (k["op"].startswith("mha_") or k["op"] in GEMM_OPS)]
        mf = [k for k in mf_all if k["ms"] >= 0.05]
        small = [k for k in mf_all if k["ms"] < 0.05 and k["op"] in GEMM_OPS]
        launch_bound_gemm = None
        if small:
            tsm = max(small, key=lambda k: k["ms"] * k["calls_per_step"])
            launch_bound_gemm = {"kernel": f"{tsm['op']}{tuple(tsm['dims'])}", "calls_per_step": tsm["calls_per_step"],
                                 "ms_per_launch_hip_events_eager": tsm["ms"], "tflops_at_that_time": tsm["tflops"],
                                 "ms_per_step_all_launch_bound_gemms": round(sum(k["ms"] * k["calls_per_step"] for k in small), 3),
                                 "note": "not roofline-priced: < 50 us per launch (profiles/*_summary.md has the rocprofv3 durations)"}
        # the tiled row-GEMM family as a whole (forward + input-gradient products of the linear layers; the streaming
        # kernels of the SA layers are inside sa_fused_*): priced by its dominant shape, family totals alongside
        roofline_gemm = None
        fam = [k for k in mf_all if k["op"] in GEMM_OPS]
        if fam:
            topg = max([k for k in fam if k.get("stream") == "main"] or fam, key=lambda k: k["ms"] * k["calls_per_step"])
            fam_ms = sum(k["ms"] * k["calls_per_step"] for k in fam)
            fam_fl = sum(algorithmic_flops((k["op"],) + tuple(k["dims"])) * k["calls_per_step"] for k in fam)
            roofline_gemm = {"kernel": f"{topg['op']}{tuple(topg['dims'])}", "bound": "mfma", "achieved": topg["tflops"],
                             "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(topg["tflops"] / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                             "alg_flops_per_launch": algorithmic_flops((topg["op"],) + tuple(topg["dims"])),
                             "ms_per_launch": topg["ms"], "calls_per_step": topg["calls_per_step"],
                             "ms_per_step": round(fam_ms, 4),
                             "family": "own tiled row GEMMs (csrc/gemm.hip gemm_rows_kernel / gemm_dma_kernel), forward + input gradients",
                             "family_tflops": round(fam_fl / (fam_ms * 1e-3) / 1e12, 2) if fam_ms > 0 else None,
                             "family_frac": round(fam_fl / (fam_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4) if fam_ms > 0 else None,
                             "by_shape": [{"kernel": f"{k['op']}{tuple(k['dims'])}", "calls_per_step": k["calls_per_step"],
                                           "ms": k["ms"], "tflops": k["tflops"]} for k in
                                          sorted(fam, key=lambda k: -k["ms"] * k["calls_per_step"])[:6]],
                             "timing": "HIP events around each launch in eager runs (for the < 20 us launches ~30 % above "
                                       "the rocprofv3 kernel durations in profiles/)",
                             "dtype": "f32 in / f32 accumulate MFMA"}
        # the dominant tiled product once more, the way the step actually runs it: 50 launches back to back in a replayed
        # hipGraph (a HIP-event bracket around ONE eager launch of a ~10 us kernel measures the bracket too: ~+35 %; the
        # rocprofv3 average in profiles/*_summary.md is the third view of the same kernel)
        # (not under a live process group: RCCL's watchdog thread queries its events while this thread captures -- "operation
        # not permitted on an event last recorded in a capturing stream" ended a --force-dist run after the line was printed)
        if roofline_gemm and topg["op"] in ("gemm_fwd", "gemm_dgrad") and not dist_on:
            try:
                from eda_amd import gemm as _g
                r_, k_, n_ = topg["dims"]
                # EIGHT distinct operand sets rotate through the 50 launches (with ONE set the operands
                # stay L2-resident, a best case the step never sees -- there every launch of this shape reads activations
                # another kernel has just written and weights it last saw a layer ago)
                NSET_ = 8
                xs_ = [torch.randn(r_, k_, device=device) for _ in range(NSET_)]
                ws_ = [torch.randn((n_, k_) if topg["op"] == "gemm_fwd" else (k_, n_), device=device) * 0.05 for _ in range(NSET_)]
                ys_ = [torch.empty(r_, n_ if topg["op"] == "gemm_fwd" else k_, device=device) for _ in range(NSET_)]
                if topg["op"] == "gemm_fwd":
                    fn_ = lambda i: _g.linear_fwd(xs_[i % NSET_], ws_[i % NSET_], out=ys_[i % NSET_])       # noqa: E731
                else:
                    fn_ = lambda i: _g.linear_dgrad(xs_[i % NSET_], ws_[i % NSET_])                          # noqa: E731
                gs_ = torch.cuda.Stream()
                gg_ = torch.cuda.CUDAGraph()
                with torch.cuda.stream(gs_), torch.no_grad():
                    for i_ in range(NSET_):
                        fn_(i_)
                    torch.cuda.synchronize()
                    with torch.cuda.graph(gg_, stream=gs_, capture_error_mode="thread_local"):
                        for i_ in range(50):
                            fn_(i_)
                for _ in range(3):
                    gg_.replay()
                torch.cuda.synchronize()
                ge0, ge1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ge0.record()
                for _ in range(20):
                    gg_.replay()
                ge1.record()
                torch.cuda.synchronize()
                t_ = ge0.elapsed_time(ge1) / (20 * 50)             # ms per launch
                fl_ = roofline_gemm["alg_flops_per_launch"]
                roofline_gemm["ms_per_launch_graph_replay"] = round(t_, 5)
                roofline_gemm["achieved_graph_replay"] = round(fl_ / (t_ * 1e-3) / 1e12, 2)
                roofline_gemm["frac_graph_replay"] = round(fl_ / (t_ * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)
                del gg_
            except Exception as exc:      # a measurement aid: never lose the line to it
                roofline_gemm["graph_replay_error"] = f"{type(exc).__name__}: {exc}"[:200]
        peak16 = 2500.0                    # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA
        roofline_mfma = None
        if mf:
            # dominant attention launch (the 1024x1024 point self-attention) against the fp32-MFMA peak
            top = max(mf, key=lambda k: k["ms"] * k["calls_per_step"])
            is16 = top["op"].startswith("mha_") and args.attn_dtype != "f32"
            mpeak = peak16 if is16 else MFMA_F32_PEAK_TFLOPS
            roofline_mfma = {"kernel": f"{top['op']}{tuple(top['dims'])}", "bound": "mfma",
                             "achieved": top["tflops"], "peak": mpeak, "unit": "TFLOP/s",
                             "frac": round(top["tflops"] / mpeak, 4),
                             "traffic": pmc_traffic.get((top["op"], tuple(top["dims"]))),
                             "traffic_source": traffic_how,
                             "alg_flops_per_launch": algorithmic_flops((top["op"],) + tuple(top["dims"])),
                             "ms_per_launch": top["ms"],
                             "dtype": ("%s in / f32 accumulate MFMA (head_dim 36 padded to 48: padding not counted)" % args.attn_dtype)
                             if is16 else "f32 in / f32 accumulate MFMA",
                             "ms_per_step": round(top["ms"] * top["calls_per_step"], 4)}
        # ---- `roofline`: ONE named kernel, chosen by a rule that is FIXED from round 5 on ----------
        # Candidates: single-shape native launches on the step's MAIN stream (the critical path) that a roof prices --
        # MFMA: mha_fwd / mha_bwd / gemm_fwd / gemm_dgrad / linear_add_dropout_ln_fwd; HBM: launches that move >= 32 MB.
        # Multi-kernel calls (sa_fused_*, wgrad_grouped) and the latency-bound sampler are not candidates (they are in
        # `kernels` / `fps`).  Winner: the largest ms_per_step = calls_per_step x ms_per_launch, ms_per_launch from the
        # HIP-event brackets of the eager kernel-timing runs.  `ms_per_step` is THAT shape's time; `family_ms_per_step`
        # its family's.  For a winner shorter than 50 us per launch the HIP-event bracket is mostly the bracket, so
        # `achieved` / `frac` are then priced on `ms_per_launch_graph_replay` (50 launches back to back in a replayed
        # hipGraph -- what the step's own graph does; profiles/r05_gemm_shapes.txt holds the rocprofv3 average of the same
        # kernel and shape) and the eager number stays beside it.
        def _cand(k):
            if k.get("stream") != "main" or k["op"] == "furthest_point_sampling":
                return False
            return bool(k["tflops"] and (k["op"].startswith("mha_") or k["op"] in GEMM_OPS)) or \
                (k["design_bytes"] >= (32 << 20) and not k["op"].startswith(("sa_fused", "wgrad")))
        cands = [k for k in kernels if _cand(k)]
        roofline = None
        if cands:
            win = max(cands, key=lambda k: k["ms"] * k["calls_per_step"])
            key = (win["op"], tuple(win["dims"]))
            is_mfma = bool(win["tflops"])
            fam_pred = (lambda k: k["op"].startswith("mha_")) if win["op"].startswith("mha_") else \
                (lambda k: k["op"] in GEMM_OPS) if win["op"] in GEMM_OPS else (lambda k: k["op"] == win["op"])
            flops_ = algorithmic_flops(key[:1] + key[1]) if is_mfma else None
            roofline = {"kernel": f"{win['op']}{tuple(win['dims'])}", "bound": "mfma" if is_mfma else "hbm",
                        "selection": "fixed rule (round 5 on): roof-priced single-shape launch with the most main-stream "
                                     "(critical-path) time per step; text-encoder launches run on the second stream and are "
                                     "excluded",
                        "peak": MFMA_F32_PEAK_TFLOPS if is_mfma else HBM_PEAK_GBS, "unit": "TFLOP/s" if is_mfma else "GB/s",
                        "calls_per_step": win["calls_per_step"], "ms_per_launch_hip_events_eager": win["ms"],
                        "ms_per_launch": win["ms"], "ms_per_step": round(win["ms"] * win["calls_per_step"], 4),
                        "family_ms_per_step": round(sum(k["ms"] * k["calls_per_step"] for k in kernels
                                                        if fam_pred(k) and k.get("stream") == "main"), 4),
                        "traffic": pmc_traffic.get(key), "traffic_source": traffic_how,
                        "dtype": "f32 in / f32 accumulate MFMA" if is_mfma else "f32"}
            if is_mfma:
                roofline["alg_flops_per_launch"] = flops_
                if roofline_gemm and roofline_gemm["kernel"] == roofline["kernel"] and roofline_gemm.get("ms_per_launch_graph_replay") \
                        and win["ms"] < 0.05:
                    t_ = roofline_gemm["ms_per_launch_graph_replay"]
                    roofline["ms_per_launch"] = t_
                    roofline["ms_per_launch_graph_replay"] = t_
                    roofline["ms_per_step"] = round(t_ * win["calls_per_step"], 4)
                    roofline["timing"] = ("50 launches back to back in a replayed hipGraph, 8 distinct operand sets in rotation "
                                          "(HIP events around the replays)")
                else:
                    roofline["timing"] = "HIP events around each launch in the eager kernel-timing runs"
                roofline["achieved"] = round(flops_ / (roofline["ms_per_launch"] * 1e-3) / 1e12, 2)
                roofline["frac"] = round(roofline["achieved"] / MFMA_F32_PEAK_TFLOPS, 4)
                # beside the guide's peak: the rate v_mfma_f32_16x16x4_f32 really issues at with the chip busy (one per 36
                # cycles per SIMD at 2.17 GHz, tools/probe/mfma_f32_rate.hip, profiles/r06_mha4_keys_per_wave.md)
                roofline["peak_measured_issue_rate"] = MFMA_F32_MEASURED_TFLOPS
                roofline["frac_of_measured_issue_rate"] = round(roofline["achieved"] / MFMA_F32_MEASURED_TFLOPS, 4)
                fam_k = [k for k in kernels if fam_pred(k) and k.get("stream") == "main" and k["tflops"]]
                fam_ms = sum(k["ms"] * k["calls_per_step"] for k in fam_k)
                fam_fl = sum(algorithmic_flops((k["op"],) + tuple(k["dims"])) * k["calls_per_step"] for k in fam_k)
                roofline["family_frac"] = round(fam_fl / (fam_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4) if fam_ms > 0 else None
            else:
                roofline["achieved"] = win["gbs"]
                roofline["frac"] = round(win["gbs"] / HBM_PEAK_GBS, 5)
                roofline["alg_bytes_per_launch"] = win["alg_bytes"]
                roofline["frac_algorithmic"] = round(win["alg_bytes"] / (win["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                roofline["timing"] = "HIP events around each launch in the eager kernel-timing runs"
        if roofline_mfma:
            roofline_mfma["family_ms_per_step"] = round(sum(k["ms"] * k["calls_per_step"] for k in kernels
                                                            if k["op"].startswith("mha_")), 4)
        fps = [k for k in kernels if k["op"] == "furthest_point_sampling"]
        fps_info = [{"n": k["dims"][1], "m": k["dims"][2], "ms": k["ms"],
                     "us_per_round": round(k["ms"] * 1e3 / max(1, k["dims"][2] - 1), 3)} for k in fps]
        out = {
            "metric": "scenes/sec fwd+bwd (50k pts, 256 queries, 80 tok)",
            "value": round(scenes / dt, 3), "unit": "scenes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.attn_dtype == "f32" else "f32 (attention QK^T/PV contractions in %s)" % args.attn_dtype,
            "data": "synthetic" + (" (%d batches resident in HBM, trained in turn)" % nb if nb > 1 else " (one batch)"),
            "config": {"workload": "BeaUTyDETR train step (fwd+bwd+clip+AdamW), butd=%s, loss=%s" % (
                           not args.no_butd, args.loss),
                       "scenes_per_gpu": args.per_gpu, "global_batch": args.per_gpu * world,
                       "points": args.points, "queries": args.queries, "tokens": args.tokens,
                       "parallelism": f"dp{world}" + (" (--force-dist: one-rank RCCL group, the N > 1 step structure)" if (args.force_dist and world == 1) else "") + (" (ALL RANKS ON ONE GPU, gloo collectives: functional run of the N > 1 "
                                                       "structure, not a measurement)" if args.share_gpu else ""),
                       "gradient_allreduce": (None if not dist_on else
                                              "two ranges of the flat fp32 buffer, the first in flight underneath the second "
                                              "range's grouped weight-gradient kernel" if overlap_ar else
                                              "one all-reduce of the flat fp32 buffer after the backward"),
                       "batchnorm": ("global-batch statistics (sync_bn, %s%s%s)" % (
                           args.sync_bn, "; one rank: the N > 1 code path" if world == 1 else "",
                           ("; in-kernel exchange, slab memory kind %d, self-test passed, 0 timed-out polls"
                            % _lib_peer_kind()) if sync_bn_native_on() else
                           ("; FELL BACK to collectives: the peer-memory self-test failed" if args.sync_bn == "native" else "")))
                       if (args.sync_bn and dist_on) else "per-GPU statistics",
                       "launch": ("eager" if not args.graph else
                                  ("three hipGraphs on two streams (frozen text encoder underneath the point backbone | "
                                   "rest of the step)" + ("" if world == 1 and not args.split_graphs else
                                                          " + clip/AdamW graph behind the RCCL all-reduce"))
                                  if (args.text_stream and not args.overlap) else
                                  "hipGraph replay of the whole step" if world == 1 and not args.split_graphs else
                                  "two hipGraphs (fwd+bwd | clip+AdamW) with the RCCL all-reduce between them"),
                       "text_encoder": "RoBERTa-base random-init frozen, " + (
                           "in train mode like the rest of the model (main_utils.py:459: its dropout layers are active), "
                           if args.text_encoder_mode == "train" else "in eval mode (no dropout inside it), ") + (
                           "forward on own kernels (eda_amd/roberta_fast.py)" if fast_roberta else "stock Hugging Face forward (hipBLASLt / AOTriton)") + (
                           "; runs for the NEXT step's tokens on the second stream" if (args.graph and args.text_stream and not args.overlap
                                                                                      and args.fps_prefetch and args.text_prefetch) else ""),
                       "sa1_sampling": (("furthest point sampling" if args.fps_prefetch == 1 else
                                         "coordinate-only geometry (4 samplings, 4 ball queries, 2 3-NN searches)") +
                                        " of the NEXT step's batch on the second stream during the "
                                        "current step (once per step; --fps-prefetch 0 puts it back on the critical path)")
                       if (args.graph and args.text_stream and not args.overlap and args.fps_prefetch) else "inside the step",
                       "optimizer": ("clip_grad_norm_(0.1) + AdamW(3 lr groups, weight_decay 5e-4) as two launches on the flat buffers "
                                     "(csrc/optim.hip; torch.optim.AdamW keeps the state)") if flat_opt is not None else
                                    "FlatParams.clip_grad_norm_ + torch.optim.AdamW(fused, capturable)",
                       "attention_dtype": args.attn_dtype,
                       "deterministic": bool(args.deterministic),
                       "own_gemms": "every pointwise layer of the model AND of the frozen text encoder (csrc/gemm.hip)" if fast_roberta
                       else "every pointwise layer of the model (csrc/gemm.hip); hipBLASLt only inside RoBERTa",
                       "roberta_gemm_selection": ("TunableOp (%s; shipped results %s)" % (
                           args.gemm_tuning, "loaded" if shipped_ok else "not used"))
                       if args.gemm_tuning != "off" else "library default"},
            "roofline": roofline,
            "launch_bound_gemm": launch_bound_gemm,
            "roofline_hbm": roofline_hbm,
            "roofline_hbm_eval": roofline_hbm_eval,
            "roofline_mfma": roofline_mfma,
            "roofline_gemm": roofline_gemm,
            "in_step": in_step,
            "native_ms_per_step": round(native_ms, 3),
            "step_matrix": {"gflop_per_step": round((mat["main"] + mat["side"]) / 1e9, 1), "gflop_main_stream": round(mat["main"] / 1e9, 1),
                            "gflop_second_stream": round(mat["side"] / 1e9, 1),
                            "tflops_over_the_step": round((mat["main"] + mat["side"]) / (ms_per_step * 1e-3) / 1e12, 2),
                            "frac_of_nominal_fp32_mfma": round((mat["main"] + mat["side"]) / (ms_per_step * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                            "frac_of_measured_issue_rate": round((mat["main"] + mat["side"]) / (ms_per_step * 1e-3) / 1e12 / MFMA_F32_MEASURED_TFLOPS, 4),
                            "note": "algorithmic FLOPs of every product-shaped native call of one step (attention, row products, SA "
                                    "stacks, weight gradients, the frozen text encoder) over the step's wall time: the share of the fp32 "
                                    "matrix pipe the WHOLE step keeps busy (its bf16 x 3 kernels are counted at their fp32 FLOPs)"},
            "fps": fps_info,
            "kernel_timing": (("HIP events on the launch stream, %d eager runs of the same step after the "
                               "graph-replayed timed region" % args.kernel_steps) if args.graph else
                              "HIP events on the launch stream inside the timed region") +
                             "; per op and shape the median bracket (mean below five calls)",
            "kernels": kernels[:40],
            "gemm_family_ms_per_step": round(sum(k["ms"] * k["calls_per_step"] for k in kernels
                                                 if k["op"].startswith(("gemm_", "sa_fused", "wgrad"))), 3),
            "source_hash": source_hash(),
            "loss": final_loss,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = run_cpu_baseline(args)
        if os.environ.get("EDA_BENCH_KERNELS_FILE"):       # the complete per-op table (the line carries the first 40 rows)
            with open(os.environ["EDA_BENCH_KERNELS_FILE"], "w") as f:
                json.dump(kernels, f)
        # the JSON line is the LAST line of stdout: RCCL prints a version banner through C stdio (buffered until exit when
        # stdout is a pipe) -- flush it out first
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
