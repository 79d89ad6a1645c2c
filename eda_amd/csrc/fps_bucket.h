// fps_bucket.h -- launch interface of the single-workgroup bucket sampler (csrc/fps_bucket.hip) under
// eda_furthest_point_sampling_f32 (csrc/fps.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// scenes of 8193..65536 points
bool eda_fps_bucket_supports(int n);
// sorted points of every scene: x | y | z | running distance | tie key, n rounded up to 64 each
size_t eda_fps_bucket_workspace_bytes(int b, int n);
// status: the 64-int status block of the FPS workspace (diagnostics: ints 3..; int 2 counts recovered give-ups).
// only_if: NULL, or a device flag -- the launch is a no-op unless *only_if != 0 (fallback behind the cluster kernels).
int eda_fps_bucket_launch(const float *xyz, int b, int n, int m, int *idx, int p_log2, void *ws, int *status,
                          const int *only_if, int mode, hipStream_t stream);
