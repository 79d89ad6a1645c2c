// eda_common.h -- shared host/device helpers for libeda_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/eda_hip.h"

#define EDA_WAVE 64

// ---- host-side error plumbing ------------------------------------------
void eda_set_error(const char *fmt, ...);
extern int g_eda_fma_mode;

// ---- the EDA_* environment knobs ------------------------------------------
// Every tuning / debugging switch of the library is one row of this table (capi.hip holds names and defaults).  The
// environment is read ONCE, on first use, into one process-wide EdaEnv; nothing in a launch path calls getenv().
// eda_reload_env() (C ABI) re-reads it and resets the setter-backed state to "from the environment" -- for tests that
// flip a knob inside a process.  eda_env_epoch() changes with every reload: code that caches a value DERIVED from a
// knob compares it.
enum EdaKnob : int {
  EDA_K_FPS_CU_RESERVE, EDA_K_FPS_BUCKET, EDA_K_FPS_BUCKET_NW, EDA_K_FPS_SMALL_T, EDA_K_FPS_SPEC, EDA_K_FPS_T, EDA_K_FPS_P,
  EDA_K_FPS_TEST_GIVEUP, EDA_K_BQ_SCAN, EDA_K_GEMM_DBG, EDA_K_GEMM_DMA, EDA_K_GEMM_DMA_MAP, EDA_K_GEMM_STREAM_GRID,
  EDA_K_GEMM_STREAM_B3, EDA_K_GEMM_STREAM, EDA_K_GEMM_STREAM_MINR, EDA_K_GEMM_CFG, EDA_K_GEMM_LN_BM, EDA_K_GEMM_LN_VAR,
  EDA_K_GEMM_SPLITK, EDA_K_GEMM_KC96, EDA_K_GEMM_B3ROWS, EDA_K_FPS_BACKGROUND, EDA_K_MHA2_PRIO, EDA_K_MHA2_BWD_DBUF, EDA_K_MHA2_BWD_MERGE, EDA_K_MHA2_KSPLIT, EDA_K_MHA4, EDA_K_MHA3, EDA_K_MHA3_DBG, EDA_K_BN_SMALL_CQ, EDA_K_SA_LAYER_FUSE, EDA_K_SA_BNBWD_FUSE,
  EDA_K_SA_BWD_B3, EDA_K_WGRAD_BF16X3, EDA_K_WGRAD_WGS, EDA_K_DETERMINISTIC, EDA_K_FROZEN_NW, EDA_K_PEER_SPIN_LOG2, EDA_K_PEER_ALLOC, EDA_K_COUNT
};
struct EdaEnv {
  long val[EDA_K_COUNT];          // the variable's integer value, or the table's default when it is unset / empty
  bool set[EDA_K_COUNT];          // the variable is present and non-empty
  int skip_k, skip_n, skip_e;     // EDA_GEMM_STREAM_SKIP="K,N,epi" (debugging aid); skip_on = it parsed
  bool skip_on;
};
const EdaEnv &eda_env();
unsigned eda_env_epoch();
static inline long eda_knob(EdaKnob k) { return eda_env().val[k]; }
static inline bool eda_knob_set(EdaKnob k) { return eda_env().set[k]; }
extern int g_eda_deterministic;   // -1: from EDA_DETERMINISTIC
static inline bool eda_deterministic() { return (g_eda_deterministic < 0 ? eda_knob(EDA_K_DETERMINISTIC) : g_eda_deterministic) != 0; }

#define EDA_CHECK_ARG(cond, msg)                         \
  do {                                                   \
    if (!(cond)) {                                       \
      eda_set_error("%s: %s", __func__, msg);            \
      return EDA_ERR_INVALID_ARG;                        \
    }                                                    \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize for a kernel that needs more than 64 KB of dynamic LDS: set once per (kernel,
// device) -- the attribute is per device, and setting it before EVERY launch costs an eager (un-captured) caller a host call
// per launch.  Returns hipSuccess or the error of the runtime call.
#include <mutex>
#include <set>
#include <utility>
static inline hipError_t eda_set_max_dynamic_lds(const void *kern, size_t bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void *, int>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({kern, dev})) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done.insert({kern, dev});
  return e;
}

#define EDA_CHECK_LAUNCH()                                              \
  do {                                                                  \
    hipError_t e__ = hipGetLastError();                                 \
    if (e__ != hipSuccess) {                                            \
      eda_set_error("%s: launch failed: %s", __func__,                  \
                    hipGetErrorString(e__));                            \
      return (int)e__;                                                  \
    }                                                                   \
  } while (0)

#define EDA_CHECK_HIP(expr)                                             \
  do {                                                                  \
    hipError_t e__ = (expr);                                            \
    if (e__ != hipSuccess) {                                            \
      eda_set_error("%s: %s failed: %s", __func__, #expr,               \
                    hipGetErrorString(e__));                            \
      return (int)e__;                                                  \
    }                                                                   \
  } while (0)

// ---- canonical fp32 arithmetic (DESIGN.md "Canonical arithmetic") -------
// The library is built with -ffp-contract=off: every fused op is explicit.
// MODE 0: a*a + b*b + c*c as nvcc -fmad=true / LLVM's DAG combiner contract
//         it: t = b*b; t = fma(a,a,t); t = fma(c,c,t).
// MODE 1: strict IEEE, ((a*a + b*b) + c*c).
template <int MODE>
__device__ __forceinline__ float eda_sumsq3(float a, float b, float c) {
  if (MODE == 0) {
    float t = b * b;
    t = __builtin_fmaf(a, a, t);
    t = __builtin_fmaf(c, c, t);
    return t;
  } else {
    return (a * a + b * b) + c * c;
  }
}

// ---- DPP cross-lane helpers (wave64, gfx9-family DPP controls) -----------
// dpp_ctrl encodings: quad_perm = 0x00..0xFF, row_shr:n = 0x110+n,
// row_ror:n = 0x120+n.
#define EDA_DPP_QUAD_XOR1 0xB1  // quad_perm [1,0,3,2]
#define EDA_DPP_QUAD_XOR2 0x4E  // quad_perm [2,3,0,1]
#define EDA_DPP_ROW_ROR(n) (0x120 + (n))

template <int CTRL>
__device__ __forceinline__ int eda_dpp(int v) {
  // all rows / all banks enabled, bound_ctrl irrelevant for permutations
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}

// All-reduce within each 16-lane row (result uniform per row).
__device__ __forceinline__ int eda_row_max_i32(int v) {
  v = max(v, eda_dpp<EDA_DPP_QUAD_XOR1>(v));
  v = max(v, eda_dpp<EDA_DPP_QUAD_XOR2>(v));
  v = max(v, eda_dpp<EDA_DPP_ROW_ROR(4)>(v));
  v = max(v, eda_dpp<EDA_DPP_ROW_ROR(8)>(v));
  return v;
}
__device__ __forceinline__ unsigned eda_row_min_u32(unsigned v) {
  v = min(v, (unsigned)eda_dpp<EDA_DPP_QUAD_XOR1>((int)v));
  v = min(v, (unsigned)eda_dpp<EDA_DPP_QUAD_XOR2>((int)v));
  v = min(v, (unsigned)eda_dpp<EDA_DPP_ROW_ROR(4)>((int)v));
  v = min(v, (unsigned)eda_dpp<EDA_DPP_ROW_ROR(8)>((int)v));
  return v;
}
// fp32 sum / max over the 64 lanes, every lane gets the result: four DPP steps inside the 16-lane rows, two gfx950 lane swaps
// across them -- six short VALU instructions instead of the six ds_bpermute round trips of a __shfl_xor butterfly (a dependent
// chain of LDS-crossbar latencies: what the row statistics of the LayerNorm kernels waited for).
typedef unsigned eda_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float eda_wave_sum_f32(float v) {
  v += __int_as_float(eda_dpp<EDA_DPP_QUAD_XOR1>(__float_as_int(v)));
  v += __int_as_float(eda_dpp<EDA_DPP_QUAD_XOR2>(__float_as_int(v)));
  v += __int_as_float(eda_dpp<EDA_DPP_ROW_ROR(4)>(__float_as_int(v)));
  v += __int_as_float(eda_dpp<EDA_DPP_ROW_ROR(8)>(__float_as_int(v)));
  eda_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r.x) + __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float eda_wave_max_f32(float v) {
  v = fmaxf(v, __int_as_float(eda_dpp<EDA_DPP_QUAD_XOR1>(__float_as_int(v))));
  v = fmaxf(v, __int_as_float(eda_dpp<EDA_DPP_QUAD_XOR2>(__float_as_int(v))));
  v = fmaxf(v, __int_as_float(eda_dpp<EDA_DPP_ROW_ROR(4)>(__float_as_int(v))));
  v = fmaxf(v, __int_as_float(eda_dpp<EDA_DPP_ROW_ROR(8)>(__float_as_int(v))));
  eda_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}

// Wave-wide all-reduce; the result is wave-uniform (SGPR).
__device__ __forceinline__ int eda_wave_max_i32(int v) {
  v = eda_row_max_i32(v);
  int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return max(max(a, b), max(c, d));
}
__device__ __forceinline__ unsigned eda_wave_min_u32(unsigned v) {
  v = eda_row_min_u32(v);
  unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
  unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
  return min(min(a, b), min(c, d));
}

__device__ __forceinline__ int eda_lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// ---- ordered scatter-add (scatter_det.hip; the eda_deterministic() form of every gradient scatter) --------------------
// out[b*out_sb + p*out_sp + c*out_sc] = sum over r in [0, R) with idx[b*idx_sb + r] == p, ascending r, of
//   (wgt ? wgt[b*idx_sb + r] : 1) * src[b*src_sb + (r / rdiv)*src_sr + c*src_sc]        for p < P, c < C
// (every point of `out` is written: zero where nothing refers to it)
struct EdaDetScatter {
  const int *idx; const float *wgt; long idx_sb; int R, rdiv;
  const float *src; long src_sb, src_sr, src_sc;
  float *out; long out_sb, out_sp, out_sc;
  int B, P, C;
};
int eda_det_scatter_launch(const EdaDetScatter &a, hipStream_t stream);

// __syncthreads() that also retires THIS wave's LDS-DMA (global_load_lds / buffer_load ... lds).  The compiler orders a
// DMA only against the issuing wave's own later LDS reads: its s_waitcnt vmcnt lands in front of that read -- possibly
// BEHIND the barrier the other waves rely on (mha3's loop-top barrier compiled to `s_waitcnt lgkmcnt(0); s_barrier` with
// the vmcnt(0) after it), and a wave that does not read at all never waits.  Every barrier that publishes DMA'd LDS
// data goes through this (round 5: the long-key attention forward gave 1e-4-off results once two processes shared the
// GPU, exact on an idle one).
#define EDA_SYNC_DMA() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)

// Zero-fill on the stream with a kernel (not hipMemsetAsync): memset NODES of a captured
// HIP graph were observed to race with the kernel nodes that consume the zeroed buffer
// on ROCm 7.2 (ball-query cell table corrupted under graph replay), kernel nodes are not.
int eda_zero_async(void *ptr, size_t bytes, hipStream_t stream);
