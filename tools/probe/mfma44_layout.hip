// Probe of the v_mfma_f32_4x4x1_16b_f32 operand layout on gfx950 (checked before csrc/mha2.hip relies on it):
// hypothesis D[reg i][lane 4b + j] = A[lane 4b + i] * B[lane 4b + j] for the 16 blocks b.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[threadIdx.x * 4 + i] = acc[i];
}
int main() {
  float ha[64], hb[64], hd[256], *da, *db, *dd;
  for (int i = 0; i < 64; ++i) { ha[i] = 1.f + i; hb[i] = 100.f + 3.f * i; }
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
  hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int i = 0; i < 4; ++i) {
      const int blk = lane >> 2, j = lane & 3;
      const float exp = ha[4 * blk + i] * hb[4 * blk + j];
      if (hd[lane * 4 + i] != exp) { if (bad < 8) printf("lane %d reg %d: got %g expected %g\n", lane, i, hd[lane * 4 + i], exp); ++bad; }
    }
  printf("mfma_4x4x1 layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
  return bad != 0;
}
