// mfma_f32_rate.hip -- issue rate of v_mfma_f32_16x16x4_f32 on one SIMD and the shader clock during a SHORT kernel:
// W waves per SIMD x C independent accumulator chains x N instructions; cycles by s_memtime, time by the 100 MHz wall clock.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f32_rate tools/probe/mfma_f32_rate.hip && /tmp/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int C>
__global__ void k(int n, float *sink, unsigned long long *out) {
  f32x4 acc[C];
  for (int i = 0; i < C; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x;
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  f32x4 t = acc[0];
  for (int i = 1; i < C; ++i) t += acc[i];
  asm volatile("" ::"v"(t));
  __syncthreads();                       // (the OLDEST wave has issue priority and finishes first: time the whole workgroup)
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
  if (t[0] == 123.f) sink[0] = t[1];
}

template <int C>
void run(int waves_per_simd, int n, int wgs) {
  float *sink; unsigned long long *out, h[2048];
  hipMalloc(&sink, 4); hipMalloc(&out, sizeof(h));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k<C>, dim3(wgs), dim3(256 * waves_per_simd), 0, 0, n, sink, out);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, out, sizeof(unsigned long long) * 2 * wgs, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < wgs; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
  cyc /= wgs; wall /= wgs;
  const double per = cyc / ((double)n * C * waves_per_simd);
  printf("waves/SIMD %d chains %d n %5d wgs %3d: %8.0f memtime ticks %7.2f us -> %.1f ticks per MFMA per SIMD, %.0f MHz tick rate, %.1f ns per MFMA per SIMD\n",
         waves_per_simd, C, n, wgs, cyc, wall / 100.0, per, cyc / (wall / 100.0), wall * 10.0 / ((double)n * C * waves_per_simd));
}

int main() {
  for (int wgs : {1, 256}) {
    run<1>(1, 2000, wgs);
    run<2>(1, 1000, wgs);
    run<4>(1, 500, wgs);
    run<2>(2, 100, wgs);
    run<2>(2, 1000, wgs);
    run<2>(4, 500, wgs);
  }
  return 0;
}
