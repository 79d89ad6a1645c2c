"""Phase timing of the forward attention kernel (experiments): builds libeda_hip with -DEDA_MHA_PROFILE
into /tmp, runs one shape, prints average s_memtime ticks per wave-tile of each phase."""
import ctypes
import glob
import os
import subprocess

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "eda_amd", "csrc")


def main():
    lq = int(os.environ.get("LQ", 1024))
    lk = int(os.environ.get("LK", 1024))
    p = float(os.environ.get("PDROP", 0.1))
    so = "/tmp/libeda_prof.so"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-ffp-contract=off", "-munsafe-fp-atomics", "-DEDA_MHA_PROFILE",
                           "-I" + os.path.join(ROOT, "include")] + sorted(glob.glob(CSRC + "/*.hip")) + ["-o", so])
    L = ctypes.CDLL(so)
    B, H = 8, 8
    q = torch.randn(B, lq, 288, device="cuda")
    k = torch.randn(B, lk, 288, device="cuda")
    v = torch.randn(B, lk, 288, device="cuda")
    out = torch.empty_like(q)
    lse = torch.empty(B, H, lq, device="cuda")
    seed = torch.full((1,), 0, dtype=torch.int64, device="cuda")
    P, l, i, f = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float
    L.eda_mha_fwd_f32.argtypes = [P, P, P, l, l, l, l, l, l, P, i, i, i, i, i, f, f, P, ctypes.c_uint, P, P, P]

    def run():
        rc = L.eda_mha_fwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), lq * 288, 288, lk * 288, 288,
                               lk * 288, 288, None, B, H, lq, lk, 36, 36 ** -0.5, p, seed.data_ptr(), 1,
                               out.data_ptr(), lse.data_ptr(), None)
        assert rc == 0

    buf = (ctypes.c_ulonglong * 8)()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    L.eda_mha_profile_read(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    L.eda_mha_profile_read(buf)
    n = buf[6]
    names = ["issue global loads", "K reads + S MFMAs", "V reads + softmax/dropout VALU", "PV MFMAs",
             "commit (vmcnt + LDS writes)", "barrier"]
    print(f"Lq={lq} Lk={lk} p={p}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch (instrumented); {n} wave-tiles")
    tot = 0
    for j, nm in enumerate(names):
        print(f"  {nm:34s} {buf[j] / n:9.1f} ticks/wave-tile")
        tot += buf[j] / n
    print(f"  total {tot:.1f} ticks per wave-tile; {n / 10 / 1024:.1f} wave-tiles per SIMD per launch")


if __name__ == "__main__":
    main()
