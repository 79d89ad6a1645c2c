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

// ---- the environment knobs (eda_common.h: EdaKnob / EdaEnv) ------------------------------------------------------
#include <atomic>
#include <mutex>
#include <stdlib.h>
namespace {
struct KnobRow { const char *name; long dflt; };
const KnobRow kKnobs[EDA_K_COUNT] = {
    {"EDA_FPS_CU_RESERVE", 0},      // CUs the cluster sampler leaves free (eda_fps_set_cu_reserve)
    {"EDA_FPS_BUCKET", -1},         // 0 / 1: cluster / bucket sampler for every call (unset: the policy of eda_fps_set_policy)
    {"EDA_FPS_BUCKET_NW", 16},      // waves of the bucket sampler (8 | 16)
    {"EDA_FPS_SMALL_T", 512},       // 1024: 1024-thread single-workgroup sampler for 2049..8192 points
    {"EDA_FPS_SPEC", 1},            // 0: cluster kernels without the speculative hand-off
    {"EDA_FPS_T", 512},             // threads per workgroup of the cluster kernels
    {"EDA_FPS_P", -1},              // points per thread of the cluster kernels (unset: by size)
    {"EDA_FPS_TEST_GIVEUP", 0},     // test hook: behave like a cluster launch that was not co-resident
    {"EDA_BQ_SCAN", 0},             // set: ball query by linear scan instead of the uniform grid
    {"EDA_GEMM_DBG", 0},
    {"EDA_GEMM_DMA", -1},           // eda_gemm_set_dma
    {"EDA_GEMM_DMA_MAP", -1},       // 0 / 1: tile -> XCD mapping of the DMA-staged products
    {"EDA_GEMM_STREAM_GRID", 0},    // workgroups of the streaming kernels (tests: any grid must work)
    {"EDA_GEMM_STREAM_B3", 1},      // 0: fp32-MFMA streaming kernels instead of bf16 x 3
    {"EDA_GEMM_STREAM", 1},         // 0: streaming kernels off
    {"EDA_GEMM_STREAM_MINR", 32768},
    {"EDA_GEMM_CFG", -1},           // force a tile of gemm_rows_kernel
    {"EDA_GEMM_LN_BM", 0},
    {"EDA_GEMM_LN_VAR", 0},
    {"EDA_GEMM_SPLITK", -1},        // 0: no split contraction; n >= 2: n slices for every eligible launch (unset: by shape)
    {"EDA_GEMM_KC96", -1},          // 0: no 96-wide chunks; n >= 1: tile configuration n for every eligible launch (unset: by shape)
    {"EDA_GEMM_B3ROWS", 1},         // 0: the many-row plain products stay on the fp32 matrix pipe; 1: bf16 x 3 where measured faster; 2: every eligible shape
    {"EDA_FPS_BACKGROUND", 0},      // 1: the cluster sampler polls one granule per record (eda_fps_set_background)
    {"EDA_MHA2_PRIO", 1},
    {"EDA_MHA2_BWD_DBUF", 0},
    {"EDA_MHA2_BWD_MERGE", -1},     // split backward: 0 = partials summed by a second launch, 1 = by the last arriver inside the launch (unset: by size)       // 1: the short-key backward variants double-buffer their query chunks
    {"EDA_MHA2_KSPLIT", -1},        // 0: no key-split forward; n >= 2: n key slices for every eligible launch (unset: by shape)
    {"EDA_MHA4", 0},                // 0 (default: measured no faster, profiles/r06_mha4_keys_per_wave.md): short query sets against >= 512 keys stay on mha2.hip's key-split forward; 1: <= 144 queries on mha4.hip (keys per wave); 2: <= 256
    {"EDA_MHA3", 1},                // 0: the long-key attention forward stays on mha2.hip's fp32-MFMA kernel (mha3.hip: bf16 x 3)
    {"EDA_MHA3_DBG", 0},            // ablation bits of mha3.hip (timing experiments; wrong results)
    {"EDA_BN_SMALL_CQ", 4},
    {"EDA_SA_LAYER_FUSE", 1},
    {"EDA_SA_BNBWD_FUSE", 1},
    {"EDA_SA_BWD_B3", 1},           // 0: fp32-MFMA set-abstraction backward launches instead of bf16 x 3
    {"EDA_WGRAD_BF16X3", 1},        // eda_wgrad_set_arith
    {"EDA_WGRAD_WGS", 144},
    {"EDA_DETERMINISTIC", 0},       // eda_set_deterministic
    {"EDA_FROZEN_NW", 0},           // csrc/gemm_frozen.hip: 4 / 8 waves per workgroup for every launch (0: by shape)
    {"EDA_PEER_SPIN_LOG2", 24},     // log2 of the polls an in-kernel statistics exchange waits for a peer (csrc/peer.h)
    {"EDA_PEER_ALLOC", 0},          // pins the slab's allocation kind: 0 fine-grained, 1 uncached, 2 plain (unset: the first that exports an IPC handle)
};
EdaEnv g_env;
std::atomic<int> g_env_ready{0};
std::atomic<unsigned> g_env_epoch{1};
std::mutex g_env_mu;
void env_load(EdaEnv &e) {
  for (int k = 0; k < EDA_K_COUNT; ++k) {
    const char *s = getenv(kKnobs[k].name);
    e.set[k] = s && *s;
    e.val[k] = e.set[k] ? atol(s) : kKnobs[k].dflt;
  }
  e.skip_on = false; e.skip_k = e.skip_n = 0; e.skip_e = -1;
  if (const char *sk = getenv("EDA_GEMM_STREAM_SKIP"))
    e.skip_on = sscanf(sk, "%d,%d,%d", &e.skip_k, &e.skip_n, &e.skip_e) >= 2;
}
}  // namespace
int g_eda_deterministic = -1;
const EdaEnv &eda_env() {
  if (!g_env_ready.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lock(g_env_mu);
    if (!g_env_ready.load(std::memory_order_relaxed)) { env_load(g_env); g_env_ready.store(1, std::memory_order_release); }
  }
  return g_env;
}
unsigned eda_env_epoch() { return g_env_epoch.load(std::memory_order_relaxed); }
void eda_fps_env_reset();      // fps.hip
void eda_gemm_env_reset();     // gemm.hip
void eda_wgrad_env_reset();    // wgrad.hip
extern "C" int eda_reload_env(void) {
  {
    std::lock_guard<std::mutex> lock(g_env_mu);
    env_load(g_env);
    g_env_ready.store(1, std::memory_order_release);
    g_env_epoch.fetch_add(1, std::memory_order_relaxed);
  }
  g_eda_deterministic = -1;
  eda_fps_env_reset(); eda_gemm_env_reset(); eda_wgrad_env_reset();
  return 0;
}
extern "C" int eda_set_deterministic(int on) {
  if (on != 0 && on != 1) { eda_set_error("eda_set_deterministic: 0 or 1"); return EDA_ERR_INVALID_ARG; }
  g_eda_deterministic = on;
  return 0;
}
extern "C" int eda_get_deterministic(void) { return eda_deterministic() ? 1 : 0; }

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

// ---- device copy: the ACHIEVABLE HBM denominator of the roofline numbers -------
namespace {
__global__ __launch_bounds__(256) void copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t k = i;
  // four independent 16-byte loads in flight per lane (1, 4 or 8 loads in flight, 2048 .. 2^20
  // workgroups and torch's own copy_ all measure 4.4-4.8 TB/s read+write on a 2 x 1 GiB pair)
  // non-temporal loads and stores: a streaming copy re-uses nothing, so it should not displace the
  // other lines of L2 / the Infinity Cache (and the stores need no read-for-ownership)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 *s4 = reinterpret_cast<const u32x4 *>(src);
  u32x4 *d4 = reinterpret_cast<u32x4 *>(dst);
  for (; k + 3 * stride < n16; k += 4 * stride) {
    const u32x4 a = __builtin_nontemporal_load(s4 + k), b = __builtin_nontemporal_load(s4 + k + stride);
    const u32x4 c = __builtin_nontemporal_load(s4 + k + 2 * stride), d = __builtin_nontemporal_load(s4 + k + 3 * stride);
    __builtin_nontemporal_store(a, d4 + k); __builtin_nontemporal_store(b, d4 + k + stride);
    __builtin_nontemporal_store(c, d4 + k + 2 * stride); __builtin_nontemporal_store(d, d4 + k + 3 * stride);
  }
  for (; k < n16; k += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(s4 + k), d4 + k);
}
}  // namespace

extern "C" int eda_device_copy_f32(const float *src, float *dst, size_t n, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0) return 0;
  EDA_CHECK_ARG(src && dst, "null pointer");
  EDA_CHECK_ARG((n & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0 &&
                    (reinterpret_cast<uintptr_t>(dst) & 15u) == 0,
                "n must be a multiple of 4 floats and both pointers 16-byte aligned");
  const size_t n16 = n / 4;
  size_t blocks = (n16 + 1023) / 1024;
  if (blocks > 8192) blocks = 8192;          // 32 workgroups per CU: grid-stride over the rest
  hipLaunchKernelGGL(copy_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                     reinterpret_cast<const uint4 *>(src), reinterpret_cast<uint4 *>(dst), n16);
  EDA_CHECK_LAUNCH();
  return 0;
}

// ---- a prediction head's box update (models/modules.py ClsAgnosticPredictHead: center = base_xyz + residual; the next decoder
// layer's position input is cat([center, size]) of the detached values, models/bdetr.py:300-308) as one launch --------------
namespace {
__global__ __launch_bounds__(256) void center_query_pos_kernel(const float *__restrict__ base, const float *__restrict__ res,
                                                               const float *__restrict__ size, long rows, float *__restrict__ center,
                                                               float *__restrict__ qpos) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = base[3 * r + c] + res[3 * r + c];
    center[3 * r + c] = v;
    qpos[6 * r + c] = v;
    qpos[6 * r + 3 + c] = size[3 * r + c];
  }
}
}  // namespace

// base, res, size, center: (rows, 3) dense; qpos: (rows, 6) = [center | size]
extern "C" int eda_center_query_pos_f32(const float *base, const float *res, const float *size, long rows, float *center, float *qpos,
                                        void *stream_) {
  EDA_CHECK_ARG(rows >= 0, "bad dimension");
  if (rows == 0) return 0;
  EDA_CHECK_ARG(base && res && size && center && qpos, "null pointer");
  hipLaunchKernelGGL(center_query_pos_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, base, res, size,
                     rows, center, qpos);
  EDA_CHECK_LAUNCH();
  return 0;
}

// ---- transposed copies of many small matrices in one launch --------------------------------------------
// The input gradient of a linear layer, dX = dY W with W (N, K) row-major, contracts over W's ROW index; the
// row-GEMM kernels read a weight fastest along the contraction (gemm.hip: "NT" form, 16-byte fragments).  Instead
// of a second kernel form with 4-byte fragment reads, the host keeps W^T (K, N) of every linear weight in a
// shadow buffer, refreshed by ONE launch right before a backward pass (eda_amd/wt_shadow.py), and every input
// gradient is an NT product too.  desc: count x 5 int64 {src pointer, dst pointer, rows, cols, first tile}.
namespace {
__global__ __launch_bounds__(256) void transpose_batch_kernel(const long long *__restrict__ desc, int count) {
  __shared__ float tile[32][33];
  const long long t = blockIdx.x;
  int lo = 0, hi = count - 1;                     // last matrix whose first tile is <= t
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid * 5 + 4] <= t) lo = mid; else hi = mid - 1;
  }
  const float *src = reinterpret_cast<const float *>(desc[lo * 5 + 0]);
  float *dst = reinterpret_cast<float *>(desc[lo * 5 + 1]);
  const int rows = (int)desc[lo * 5 + 2], cols = (int)desc[lo * 5 + 3];
  const int lt = (int)(t - desc[lo * 5 + 4]);
  const int tcols = (cols + 31) / 32;
  const int r0 = (lt / tcols) * 32, c0 = (lt % tcols) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 8 * i][tx] = src[(long long)r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < rows && c < cols) dst[(long long)c * rows + r] = tile[tx][ty + 8 * i];
  }
}
}  // namespace

extern "C" int eda_transpose_batch_f32(const long long *desc, int count, long long total_tiles, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (count <= 0 || total_tiles <= 0) return 0;
  EDA_CHECK_ARG(desc && total_tiles < 0x7fffffffLL, "bad descriptor table");
  hipLaunchKernelGGL(transpose_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, stream, desc, count);
  EDA_CHECK_LAUNCH();
  return 0;
}

// ---- out = sum of up to 8 equally shaped tensors, one launch -------------------------------------------------------
// A tensor with n consumers costs the autograd engine n - 1 separate accumulation launches in the backward (at::add per
// incoming gradient: 73 of them per training step, ~5 us each, for the residual streams / positional terms of the
// encoder and decoder layers).  eda_amd.nn_utils.fan_out gives every consumer its own alias and sums the incoming
// gradients HERE in one launch.  Order of the additions: ((g0 + g1) + g2) + ... (the order the arguments are given in).
namespace {
struct AddN { const float *src[8]; };
__global__ __launch_bounds__(256) void add_n_kernel(const AddN a, int n, float4 *__restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 t = reinterpret_cast<const float4 *>(a.src[0])[i];
    for (int k = 1; k < n; ++k) {
      const float4 v = reinterpret_cast<const float4 *>(a.src[k])[i];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    out[i] = t;
  }
}
}  // namespace

extern "C" int eda_add_n_f32(const float *const *srcs, int n, size_t count, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(n >= 1 && n <= 8 && srcs && out, "1..8 source tensors");
  EDA_CHECK_ARG(count % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0, "count must be a multiple of 4, 16-byte aligned");
  if (count == 0) return 0;
  AddN a;
  for (int k = 0; k < 8; ++k) {
    a.src[k] = k < n ? srcs[k] : nullptr;
    EDA_CHECK_ARG(k >= n || (srcs[k] && (reinterpret_cast<uintptr_t>(srcs[k]) & 15u) == 0), "sources must be 16-byte aligned");
  }
  size_t blocks = (count / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_n_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, n, reinterpret_cast<float4 *>(out), count / 4);
  EDA_CHECK_LAUNCH();
  return 0;
}
