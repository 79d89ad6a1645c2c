// glds_rate.hip -- how fast one CU can stage global memory into LDS with global_load_lds (16 B per lane), by address
// pattern of the 64 lanes of an instruction.  hipcc --offload-arch=gfx950 -O3 -o glds_rate glds_rate.hip && ./glds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int PAT>
__global__ __launch_bounds__(256) void k(const float *base, long region_floats, int iters, int wg_stride_kb, float *sink) {
  __shared__ __attribute__((aligned(1024))) float lds[4 * 8 * 256];          // 8 KB per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // element offset of this lane inside one 1 KB "piece" p (p counts pieces of this wave)
  long off;                      // floats
  long piece_step;               // floats between consecutive pieces
  if (PAT == 0) { off = 4 * lane; piece_step = 256; }                                       // contiguous 1 KB
  else if (PAT == 1) { off = (long)(lane >> 3) * 288 + 4 * (lane & 7); piece_step = 8 * 288; }        // 8 rows x 128 B, stride 1152 B
  else if (PAT == 2) { const int r = lane >> 3; off = (long)r * 288 + 4 * ((lane & 7) ^ ((r >> 1) & 7)); piece_step = 8 * 288; }
  else if (PAT == 3) { off = (long)(lane >> 3) * 768 + 4 * (lane & 7); piece_step = 8 * 768; }        // stride 3072 B
  else if (PAT == 4) { off = (long)(lane >> 4) * 288 + 4 * (lane & 15); piece_step = 4 * 288; }       // 4 rows x 256 B
  else { off = (long)(lane >> 2) * 288 + 4 * (lane & 3); piece_step = 16 * 288; }                     // 16 rows x 64 B
  const long wg_base = ((long)blockIdx.x * wg_stride_kb * 256) % region_floats;
  long p = wg_base + (long)wave * piece_step;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      long a = p + off;
      if (a >= region_floats - 4096) a -= region_floats - 8192 > 0 ? (region_floats - 8192) : 0;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + a),
                                       (__attribute__((address_space(3))) void *)(lds + wave * 2048 + u * 256), 16, 0, 0);
      p += 4 * piece_step;
      if (p >= region_floats - 16384) p = wg_base + (long)wave * piece_step;
    }
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));            // vmcnt(0)
  }
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = lds[1];
}

template <int PAT>
void run(const char *name, const float *d, long region_floats, float *sink, int wgs) {
  const int iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<PAT>, dim3(wgs), dim3(256), 0, 0, d, region_floats, iters, 64, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * 4 * iters * 8 * 1024.0;
  const double per_cu = bytes / (wgs < 256 ? wgs : 256) / (ms * 1e-3);
  printf("%-34s wgs %4d  region %6.1f MB  %8.3f ms  %7.1f GB/s total  %6.2f GB/s per CU  = %5.1f B/clk @2.4GHz  (%5.0f cycles per 1 KB piece per CU)\n",
         name, wgs, region_floats * 4 / 1e6, ms, bytes / (ms * 1e-3) / 1e9, per_cu / 1e9, per_cu / 2.4e9, 1024.0 / (per_cu / 2.4e9));
}

int main() {
  float *d, *sink;
  const long big = 64L << 20;           // 256 MB of floats
  hipMalloc(&d, big * 4);
  hipMemset(d, 0, big * 4);
  hipMalloc(&sink, 4096 * 4);
  for (long region : {2L << 20 >> 2, 64L << 20 >> 2, 1024L << 20 >> 2}) {       // 2 MB, 64 MB, 1 GB... capped to the buffer
    long rf = region > big ? big : region;
    for (int wgs : {256, 512, 1024}) {
      run<0>("contiguous 1 KB", d, rf, sink, wgs);
      run<1>("8 rows x 128 B, stride 1152", d, rf, sink, wgs);
      run<2>("8 rows x 128 B, XOR granules", d, rf, sink, wgs);
      run<3>("8 rows x 128 B, stride 3072", d, rf, sink, wgs);
      run<4>("4 rows x 256 B, stride 1152", d, rf, sink, wgs);
      run<5>("16 rows x 64 B, stride 1152", d, rf, sink, wgs);
    }
  }
  return 0;
}
