"""In-kernel section timing of gemm_dma_kernel (csrc/gemm.hip built with -DEDA_GEMM_PROFILE:
`python -c "from eda_amd import build; build.build_variant('gprof', ['-DEDA_GEMM_PROFILE'])"`, run with
EDA_HIP_LIB=eda_amd/csrc/libeda_hip_gprof.so and EDA_GEMM_DMA=<configuration>).  s_memtime cycles (100 MHz) per wave."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eda_amd import _lib, gemm  # noqa: E402
NAMES = ["prologue (maps, first chunks issued, landed)", "DMA issue", "multiply (ds_read + MFMA)", "wait vmcnt", "barrier", "epilogue"]
L = _lib.lib()
fn = L.eda_gemm_profile_read
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 8)()
for sh in os.environ.get("SHAPES", "640x768x2304 640x768x768 8192x288x576 2048x288x288").split():
    R, K, N = [int(v) for v in sh.split("x")]
    x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda"); y = torch.empty(R, N, device="cuda")
    for _ in range(3): gemm.linear_fwd(x, w, out=y)
    torch.cuda.synchronize(); fn(buf)
    for _ in range(5): gemm.linear_fwd(x, w, out=y)
    torch.cuda.synchronize(); fn(buf)
    waves = buf[7]; tot = sum(buf[i] for i in range(6))
    print(f"{sh} cfg {os.environ.get('EDA_GEMM_DMA')}: {waves // 5} waves per launch, {tot / waves:.0f} ticks per wave ({K // 32} chunks)")
    for i, n in enumerate(NAMES):
        print(f"   {n:48s} {buf[i] / waves:9.1f}  {100.0 * buf[i] / tot:5.1f} %")
