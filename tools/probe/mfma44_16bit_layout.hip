// Probe of v_mfma_f32_4x4x4_16b_f16 / _bf16 (16 blocks of 4x4x4) on gfx950, before csrc/mha2.hip relies on it:
// hypothesis D[reg i][lane 4b + j] = sum_k A[lane 4b + i][k] * B[lane 4b + j][k], k = 0..3 = the lane's four packed values.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d, float* e) {
  const int l = threadIdx.x;
  h16x4 ha, hb; s16x4 sa, sb;
  for (int i = 0; i < 4; ++i) {
    ha[i] = (_Float16)a[4 * l + i]; hb[i] = (_Float16)b[4 * l + i];
    sa[i] = (short)(__float_as_uint(a[4 * l + i]) >> 16); sb[i] = (short)(__float_as_uint(b[4 * l + i]) >> 16);   // exact: small integers
  }
  f32x4 z = {0, 0, 0, 0};
  f32x4 r1 = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, z, 0, 0, 0);
  f32x4 r2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(sa, sb, z, 0, 0, 0);
  for (int i = 0; i < 4; ++i) { d[l * 4 + i] = r1[i]; e[l * 4 + i] = r2[i]; }
}
int main() {
  float ha[256], hb[256], hd[256], he[256], *da, *db, *dd, *de;
  for (int i = 0; i < 256; ++i) { ha[i] = (float)(1 + (i * 7) % 13); hb[i] = (float)(2 + (i * 5) % 11); }
  hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 1024); hipMalloc(&de, 1024);
  hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd, de);
  hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost); hipMemcpy(he, de, 1024, hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int i = 0; i < 4; ++i) {
      const int blk = lane >> 2, j = lane & 3;
      float exp = 0;
      for (int kk = 0; kk < 4; ++kk) exp += ha[4 * (4 * blk + i) + kk] * hb[4 * (4 * blk + j) + kk];
      if (hd[lane * 4 + i] != exp) ++bad1;
      if (he[lane * 4 + i] != exp) ++bad2;
    }
  printf("mfma_4x4x4 f16 layout hypothesis: %s (%d mismatches); bf16: %s (%d mismatches)\n", bad1 ? "WRONG" : "confirmed", bad1,
         bad2 ? "WRONG" : "confirmed", bad2);
  return (bad1 || bad2) != 0;
}
