// capi.hip -- library-level entry points of libeda_hip.so (include/eda_hip.h).
#include "eda_common.h"

#include <stdarg.h>

int g_eda_fma_mode = 0;

static thread_local char g_err[512] = "";

void eda_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int eda_version(void) { return EDA_HIP_ABI_VERSION; }
extern "C" const char *eda_last_error_string(void) { return g_err; }
extern "C" int eda_set_fma_mode(int mode) {
  if (mode != 0 && mode != 1) {
    eda_set_error("eda_set_fma_mode: mode must be 0 or 1");
    return EDA_ERR_INVALID_ARG;
  }
  g_eda_fma_mode = mode;
  return 0;
}
extern "C" int eda_get_fma_mode(void) { return g_eda_fma_mode; }

namespace {
__global__ __launch_bounds__(256) void zero_kernel(uint4 *p16, size_t n16, unsigned char *tail, size_t ntail) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t k = i; k < n16; k += stride) p16[k] = make_uint4(0u, 0u, 0u, 0u);
  if (i < ntail) tail[i] = 0;
}
}  // namespace

int eda_zero_async(void *ptr, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return 0;
  unsigned char *p = reinterpret_cast<unsigned char *>(ptr);
  size_t head = (16 - (reinterpret_cast<uintptr_t>(p) & 15u)) & 15u;
  if (head > bytes) head = bytes;
  if (head) {   // unaligned prefix (never the case for torch allocations): one tiny launch
    hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(256), 0, stream, nullptr, (size_t)0, p, head);
    p += head; bytes -= head;
  }
  const size_t n16 = bytes / 16, ntail = bytes % 16;
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                     reinterpret_cast<uint4 *>(p), n16, p + n16 * 16, ntail);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { eda_set_error("eda_zero_async: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}
