"""Hand-offs the speculative FPS kernel needed per scene (status ints 4.. of its workspace) and
its time, called straight through the C ABI:  python tools/fps_handoffs.py [B] [N] [M]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eda_amd import _lib, synthetic  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    L = _lib.lib()
    xyz = torch.from_numpy(synthetic.batch(range(B), N)[:, :, :3].copy()).cuda().contiguous()
    idx = torch.empty((B, M), dtype=torch.int32, device="cuda")
    nb = L.eda_fps_workspace_bytes(B, N, M)
    ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def run():
        rc = L.eda_furthest_point_sampling_f32(xyz.data_ptr(), B, N, M, idx.data_ptr(), ws.data_ptr(), nb, st)
        _lib.check(rc, "eda_furthest_point_sampling_f32")
    run(); run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    ho = ws[:256].view(torch.int32)[4:4 + B].tolist()
    prof = ws[:256].view(torch.int32)[16:23].tolist()
    if any(prof):
        names = ["update+lists", "wave top-K", "barrier1", "wg rank+publish", "poll", "global rank+accept", "barrier2"]
        if L.eda_fps_workspace_bytes(B, N, M) > 256 + B * 2 * 64 * 8 * 8 and os.environ.get("EDA_FPS_BUCKET", "1") != "0":
            names = ["candidates", "flag pass", "dense rounds", "load", "round update", "round barrier", "arg-max"]
            extra = ws[:256].view(torch.int32)[24:28].tolist()
            print("   bucket sampler, scene 0: dense rounds %d, register rounds %d (unsafe %d), groups loaded %d" % tuple(extra))
        tot = sum(prof)
        print("   cycle shares (wave 0 of workgroup 0, scene 0): " +
              ", ".join(f"{n} {100 * v / tot:.0f}%" for n, v in zip(names, prof)) +
              f"; {16 * tot / max(1, ho[0]):.0f} ticks per hand-off")
    print(f"B={B} N={N} M={M}: {ms:.3f} ms; hand-offs per scene {ho}; "
          f"{ms * 1e3 / max(1, max(ho)):.2f} us per hand-off, {ms * 1e3 / (M - 1):.2f} us per sample")


if __name__ == "__main__":
    main()
