/*
 * eda_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the nine CUDA ops of yanmin-wu/EDA's
 * pointnet2/_ext_src.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or call this library, and only as the checker.
 * The product path (eda_amd/ -> libeda_hip.so) never touches it.
 *
 * PARITY STATUS: "parity unpinned by the reference" for FPS / ball_query /
 * group / gather / three_nn -- the reference ships no CPU implementation
 * (every dispatcher ends in TORCH_CHECK(false,"CPU not supported"):
 * sampling.cpp:39,65,87  ball_query.cpp:33  group_points.cpp:36,61
 * interpolate.cpp:41,71,100) and its only op test (pointnet2_test.py:18-30,
 * a three_interpolate gradcheck at tol 1e-1) needs CUDA.  The oracle is
 * pinned instead by (1) a literal thread-level emulator of the kernels
 * (oracle/eda_emulator.c) it is cross-checked against, (2) golden vectors
 * produced by running the reference's own Python layers in the build
 * container with this oracle injected as pointnet2._ext (tests/golden/).
 *
 * All file:line citations are relative to /root/reference/pointnet2/_ext_src.
 */
#ifndef EDA_ORACLE_H
#define EDA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Arithmetic mode of the squared-distance expression
 *   (ax-bx)*(ax-bx) + (ay-by)*(ay-by) + (az-bz)*(az-bz)
 * 0 = "nvcc -fmad=true" contraction as LLVM's DAG combiner produces it:
 *       t = dy*dy; t = fmaf(dx,dx,t); t = fmaf(dz,dz,t)
 * 1 = strict IEEE, no contraction: ((dx*dx + dy*dy) + dz*dz)
 * The HIP kernels implement the same two modes; default is 0.            */
void eda_oracle_set_fma_mode(int mode);
int  eda_oracle_get_fma_mode(void);

/* Number of OpenMP threads used by the *_mt entry points (cpu_baseline). */
void eda_oracle_set_threads(int n);
int  eda_oracle_get_threads(void);

/* cuda_utils.h:20-24 */
int eda_oracle_opt_n_threads(int work_size);

/* sampling.cpp:70-91 + sampling_gpu.cu:74-178.
 * xyz (b,n,3) f32 -> idx (b,m) i32.  temp (b,n) f32 scratch, filled with
 * 1e10f here (sampling.cpp:78-80); idx is zero-filled first (:74-76).     */
void eda_oracle_furthest_point_sampling(int b, int n, int m, const float *xyz,
                                        float *temp, int *idx);

/* sampling.cpp:20-43 + sampling_gpu.cu:13-25.  points (b,c,n), idx (b,m) -> out (b,c,m) */
void eda_oracle_gather_points(int b, int c, int n, int m, const float *points,
                              const int *idx, float *out);
/* sampling.cpp:45-69 + sampling_gpu.cu:39-52.  grad_out (b,c,m) -> grad_points (b,c,n), zero-filled here */
void eda_oracle_gather_points_grad(int b, int c, int n, int m,
                                   const float *grad_out, const int *idx,
                                   float *grad_points);

/* ball_query.cpp:13-37 + ball_query_gpu.cu:14-49.
 * new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample) zero-filled here.     */
void eda_oracle_ball_query(int b, int n, int m, float radius, int nsample,
                           const float *new_xyz, const float *xyz, int *idx);

/* group_points.cpp:17-40 + group_points_gpu.cu:13-33 */
void eda_oracle_group_points(int b, int c, int n, int npoints, int nsample,
                             const float *points, const int *idx, float *out);
/* group_points.cpp:42-65 + group_points_gpu.cu:48-69 */
void eda_oracle_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                  const float *grad_out, const int *idx,
                                  float *grad_points);

/* interpolate.cpp:19-45 + interpolate_gpu.cu:14-64.
 * unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) f32, idx (b,n,3) i32    */
void eda_oracle_three_nn(int b, int n, int m, const float *unknown,
                         const float *known, float *dist2, int *idx);
/* interpolate.cpp:47-75 + interpolate_gpu.cu:77-106.  points (b,c,m) -> out (b,c,n) */
void eda_oracle_three_interpolate(int b, int c, int m, int n,
                                  const float *points, const int *idx,
                                  const float *weight, float *out);
/* interpolate.cpp:76-104 + interpolate_gpu.cu:121-148. grad_out (b,c,n) -> grad_points (b,c,m) */
void eda_oracle_three_interpolate_grad(int b, int c, int n, int m,
                                       const float *grad_out, const int *idx,
                                       const float *weight, float *grad_points);

/* Multi-threaded (OpenMP) variants of the three SA-stack ops, used only by
 * bench.py's cpu_baseline leg.  Same results as the scalar ones.          */
void eda_oracle_furthest_point_sampling_mt(int b, int n, int m, const float *xyz,
                                           float *temp, int *idx);
void eda_oracle_ball_query_mt(int b, int n, int m, float radius, int nsample,
                              const float *new_xyz, const float *xyz, int *idx);
void eda_oracle_group_points_mt(int b, int c, int n, int npoints, int nsample,
                                const float *points, const int *idx, float *out);

#ifdef __cplusplus
}
#endif
#endif
