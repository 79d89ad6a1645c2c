"""Run the SA2-shaped fused MLP (forward + backward) of tests/test_sa_fused_gpu.py::test_gathered_rows with EDA_GEMM_STREAM_B3=<mode>
and, on the second call (mode 2), compare everything the forward saved, the ReLU decisions and every gradient with the first
call (mode 1): how one decision on a tie turns rounding differences of 4e-6 into a rank-1 change of the weight gradients.
    python tools/compare_b3_runs.py 1; python tools/compare_b3_runs.py 2"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["EDA_GEMM_STREAM_MINR"] = "1"
mode = sys.argv[1]
os.environ["EDA_GEMM_STREAM_B3"] = mode
import test_sa_fused_gpu as T
from eda_amd import pointnet2_utils as PU
dev = "cuda"
B, N, m, ns, C, chans = 2, 2048, 256, 32, 128, [128, 128, 256]
torch.manual_seed(N + 3 * C)
rng = np.random.default_rng(N + C)
xyz = torch.from_numpy(rng.uniform(-2, 2, (B, N, 3)).astype(np.float32)).to(dev)
new_xyz = xyz[:, :m].contiguous()
idx = PU.ball_query(0.6, ns, xyz, new_xyz)
chans = [3 + C] + chans
Ws, gammas, betas, running = T._build(chans, dev, N + C + 1)
feats_cl = torch.randn(B, N, C, device=dev)
leaves = [feats_cl] + Ws + gammas + betas
for t in leaves:
    t.requires_grad_(True)
out = T._run_fused(dict(radius=0.6, normalize_xyz=True), Ws, gammas, betas, [(a.clone(), b.clone()) for a, b in running], True, ns,
                   xyz=xyz, new_xyz=new_xyz, feats_cl=feats_cl, idx=idx)
saved = out.grad_fn.saved_tensors
L = 3
z = [t.clone() for t in saved[6 + 2 * L:6 + 3 * L]]
stats = [t.clone() for t in saved[6 + 3 * L:6 + 4 * L]]
w = torch.randn(out.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
got = torch.autograd.grad((out * w).sum(), leaves)
allsaved = [t.clone().cpu() if torch.is_tensor(t) else None for t in saved]
torch.save({"saved": allsaved, "out": out.detach().cpu(), "z": [t.cpu() for t in z], "stats": [t.cpu() for t in stats], "g": [t.cpu() for t in got]}, f"/tmp/dbg_{mode}.pt")
if mode == "2":
    a = torch.load("/tmp/dbg_1.pt"); b = torch.load("/tmp/dbg_2.pt")
    print("out maxdiff", (a["out"] - b["out"]).abs().max().item())
    for l in range(3):
        d = (a["z"][l] - b["z"][l]).abs()
        rows = (d.max(dim=1)[0] > 1e-3).nonzero().flatten()
        print("layer", l, "z maxdiff", d.max().item(), "bad rows", rows.numel(), rows[:20].tolist(), " stats maxdiff", (a["stats"][l] - b["stats"][l]).abs().max().item())
        if rows.numel():
            r = rows[0].item(); cols = (d[r] > 1e-3).nonzero().flatten()
            print("   row", r, "bad cols", cols.numel(), cols[:40].tolist(), a["z"][l][r][cols[:4]].tolist(), b["z"][l][r][cols[:4]].tolist())
    for l in range(2):
        za, zb = a["z"][l], b["z"][l]
        sa, sb = a["stats"][l], b["stats"][l]
        ya, yb = za * sa[2] + sa[3], zb * sb[2] + sb[3]
        flips = ((ya > 0) != (yb > 0))
        print("layer", l, "relu decision flips:", int(flips.sum()), "of", flips.numel(), "| |y| at flips:", ya[flips].abs()[:10].tolist())
        print("   elements with |y| < 1e-5:", int((ya.abs() < 1e-5).sum()), " exact zeros in z:", int((za == 0).sum()), int((zb == 0).sum()))
    for i, (x, y) in enumerate(zip(a["saved"], b["saved"])):
        if x is None: continue
        print("saved", i, tuple(x.shape), x.dtype, "maxdiff", (x.double() - y.double()).abs().max().item() if x.numel() else 0)
    for i, (x, y) in enumerate(zip(a["g"], b["g"])):
        print("grad", i, tuple(x.shape), "maxdiff", (x - y).abs().max().item(), "scale", y.abs().max().item())
