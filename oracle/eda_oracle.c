/*
 * eda_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See eda_oracle.h for scope, parity status and the citation convention
 * (file:line relative to /root/reference/pointnet2/_ext_src).
 *
 * Build: gcc -O2 -fPIC -shared -fopenmp -ffp-contract=off (oracle/Makefile).
 * -ffp-contract=off matters: every fused multiply-add below is explicit.
 */
#include "eda_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_fma_mode = 0;
static int g_threads = 1;

void eda_oracle_set_fma_mode(int mode) { g_fma_mode = mode ? 1 : 0; }
int eda_oracle_get_fma_mode(void) { return g_fma_mode; }
void eda_oracle_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int eda_oracle_get_threads(void) { return g_threads; }

/* Hot loops are cloned for FMA hardware; the default clone calls libm's
 * (correctly rounded) fmaf, so results are identical on any x86-64 host. */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define EDA_CLONES __attribute__((target_clones("fma", "default")))
#else
#define EDA_CLONES
#endif

/* a*a + b*b + c*c in fp32, in the selected arithmetic mode. */
static inline __attribute__((always_inline)) float sumsq3(float a, float b, float c, int mode) {
  if (mode == 0) {
    float t = b * b;
    t = fmaf(a, a, t);
    t = fmaf(c, c, t);
    return t;
  }
  return (a * a + b * b) + c * c;
}

/* cuda_utils.h:20-24 -- (int)(log(w)/log(2)) then clamp 2^p to [1,512]. */
int eda_oracle_opt_n_threads(int work_size) {
  if (work_size <= 0) return 1;
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

/* ------------------------------------------------------------------ FPS */

/* One scene.  Thread-level emulation of furthest_point_sampling_kernel<bs>
 * (sampling_gpu.cu:74-178): per-thread strided scan (:100-116), then the
 * shared-memory tree with strides bs/2..1 (:121-174), slot idx1 kept on
 * ties (__update, :64-70).                                               */
EDA_CLONES
static void fps_scene(int n, int m, const float *dataset, float *temp, int *idxs,
                      int bs, int mode, float *dists, int *dists_i) {
  if (m <= 0) return;                       /* :78 */
  int old = 0;
  idxs[0] = old;                            /* :91-92 */
  for (int j = 1; j < m; j++) {             /* :95 */
    const float x1 = dataset[old * 3 + 0];
    const float y1 = dataset[old * 3 + 1];
    const float z1 = dataset[old * 3 + 2];
    /* Per-thread running best (:96-97).  Thread tid visits k = tid, tid+bs, ...
     * in ascending order (:101); walking k = 0..n-1 and updating slot
     * k mod bs visits every thread's points in that same order.          */
    for (int tid = 0; tid < bs; tid++) { dists[tid] = -1.f; dists_i[tid] = 0; }
    for (int k = 0; k < n; k++) {
      const int tid = k & (bs - 1);         /* bs is a power of two */
      const float x2 = dataset[k * 3 + 0];
      const float y2 = dataset[k * 3 + 1];
      const float z2 = dataset[k * 3 + 2];
      const float mag = sumsq3(x2, y2, z2, mode);       /* :106 */
      if ((double)mag <= 1e-3) continue;                /* :107, double literal */
      const float d = sumsq3(x2 - x1, y2 - y1, z2 - z1, mode); /* :109-110 point minus centre */
      const float tk = temp[k];
      const float d2 = d < tk ? d : tk;                 /* :112 min(d,temp[k]) */
      temp[k] = d2;                                     /* :113 */
      if (d2 > dists[tid]) { dists_i[tid] = k; dists[tid] = d2; }  /* :114-115, then :117-118 */
    }
    for (int s = 256; s >= 1; s >>= 1) {    /* :121-174 */
      if (bs >= 2 * s) {
        for (int tid = 0; tid < s; tid++) {
          const float v1 = dists[tid], v2 = dists[tid + s];
          const int i1 = dists_i[tid], i2 = dists_i[tid + s];
          dists[tid] = v1 > v2 ? v1 : v2;   /* max(v1,v2): equal -> either */
          dists_i[tid] = v2 > v1 ? i2 : i1; /* :69 */
        }
      }
    }
    old = dists_i[0];                       /* :176 */
    idxs[j] = old;                          /* :177 */
  }
}

static void fps_impl(int b, int n, int m, const float *xyz, float *temp, int *idx,
                     int parallel) {
  const int bs = eda_oracle_opt_n_threads(n);   /* sampling_gpu.cu:183 */
  const int mode = g_fma_mode;
  if (m > 0) memset(idx, 0, sizeof(int) * (size_t)b * m);        /* sampling.cpp:74-76 */
  for (size_t i = 0; i < (size_t)b * n; i++) temp[i] = 1e10f;    /* sampling.cpp:78-80 */
  if (n <= 0) return;
#pragma omp parallel for if (parallel) num_threads(g_threads) schedule(dynamic, 1)
  for (int i = 0; i < b; i++) {
    float dists[512];
    int dists_i[512];
    fps_scene(n, m, xyz + (size_t)i * n * 3, temp + (size_t)i * n,
              idx + (size_t)i * m, bs, mode, dists, dists_i);
  }
}

void eda_oracle_furthest_point_sampling(int b, int n, int m, const float *xyz,
                                        float *temp, int *idx) {
  fps_impl(b, n, m, xyz, temp, idx, 0);
}
void eda_oracle_furthest_point_sampling_mt(int b, int n, int m, const float *xyz,
                                           float *temp, int *idx) {
  fps_impl(b, n, m, xyz, temp, idx, 1);
}

/* --------------------------------------------------------------- gather */

void eda_oracle_gather_points(int b, int c, int n, int m, const float *points,
                              const int *idx, float *out) {
  for (int i = 0; i < b; i++)
    for (int l = 0; l < c; l++)
      for (int j = 0; j < m; j++) {
        const int a = idx[(size_t)i * m + j];                               /* :20 */
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + a]; /* :21 */
      }
}

void eda_oracle_gather_points_grad(int b, int c, int n, int m,
                                   const float *grad_out, const int *idx,
                                   float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);  /* sampling.cpp:56-58 */
  for (int i = 0; i < b; i++)
    for (int l = 0; l < c; l++)
      for (int j = 0; j < m; j++) {         /* atomicAdd order unspecified; ascending j here */
        const int a = idx[(size_t)i * m + j];
        grad_points[((size_t)i * c + l) * n + a] += grad_out[((size_t)i * c + l) * m + j];
      }
}

/* ----------------------------------------------------------- ball query */

EDA_CLONES
static void ball_query_centres(int n, int nsample, float radius2, int mode,
                               const float *new_xyz, const float *xyz, int *idx,
                               int j0, int j1) {
  for (int j = j0; j < j1; j++) {           /* ball_query_gpu.cu:28 */
    const float new_x = new_xyz[j * 3 + 0];
    const float new_y = new_xyz[j * 3 + 1];
    const float new_z = new_xyz[j * 3 + 2];
    int cnt = 0;
    for (int k = 0; k < n && cnt < nsample; ++k) {   /* :32 */
      const float x = xyz[k * 3 + 0];
      const float y = xyz[k * 3 + 1];
      const float z = xyz[k * 3 + 2];
      const float d2 = sumsq3(new_x - x, new_y - y, new_z - z, mode); /* :36-37 centre minus point */
      if (d2 < radius2) {                   /* :38 strict */
        if (cnt == 0)                       /* :39-43 pre-fill row with first hit */
          for (int l = 0; l < nsample; ++l) idx[(size_t)j * nsample + l] = k;
        idx[(size_t)j * nsample + cnt] = k; /* :44 */
        ++cnt;
      }
    }
  }
}

static void ball_query_impl(int b, int n, int m, float radius, int nsample,
                            const float *new_xyz, const float *xyz, int *idx,
                            int parallel) {
  const float radius2 = radius * radius;    /* :26 fp32 */
  const int mode = g_fma_mode;
  memset(idx, 0, sizeof(int) * (size_t)b * m * nsample);   /* ball_query.cpp:24-26 */
  const int chunk = 16;
  const int nchunks = (m + chunk - 1) / chunk;
#pragma omp parallel for if (parallel) num_threads(g_threads) schedule(dynamic, 4) collapse(2)
  for (int i = 0; i < b; i++)
    for (int cidx = 0; cidx < nchunks; cidx++) {
      const int j0 = cidx * chunk;
      const int j1 = j0 + chunk < m ? j0 + chunk : m;
      ball_query_centres(n, nsample, radius2, mode, new_xyz + (size_t)i * m * 3,
                         xyz + (size_t)i * n * 3, idx + (size_t)i * m * nsample, j0, j1);
    }
}

void eda_oracle_ball_query(int b, int n, int m, float radius, int nsample,
                           const float *new_xyz, const float *xyz, int *idx) {
  ball_query_impl(b, n, m, radius, nsample, new_xyz, xyz, idx, 0);
}
void eda_oracle_ball_query_mt(int b, int n, int m, float radius, int nsample,
                              const float *new_xyz, const float *xyz, int *idx) {
  ball_query_impl(b, n, m, radius, nsample, new_xyz, xyz, idx, 1);
}

/* ---------------------------------------------------------------- group */

static void group_points_impl(int b, int c, int n, int npoints, int nsample,
                              const float *points, const int *idx, float *out,
                              int parallel) {
#pragma omp parallel for if (parallel) num_threads(g_threads) schedule(static) collapse(2)
  for (int i = 0; i < b; i++)
    for (int l = 0; l < c; l++) {
      const float *p = points + ((size_t)i * c + l) * n;
      const int *ix = idx + (size_t)i * npoints * nsample;
      float *o = out + ((size_t)i * c + l) * npoints * nsample;
      for (int j = 0; j < npoints; j++)
        for (int k = 0; k < nsample; ++k)   /* group_points_gpu.cu:28-31 */
          o[(size_t)j * nsample + k] = p[ix[(size_t)j * nsample + k]];
    }
}

void eda_oracle_group_points(int b, int c, int n, int npoints, int nsample,
                             const float *points, const int *idx, float *out) {
  group_points_impl(b, c, n, npoints, nsample, points, idx, out, 0);
}
void eda_oracle_group_points_mt(int b, int c, int n, int npoints, int nsample,
                                const float *points, const int *idx, float *out) {
  group_points_impl(b, c, n, npoints, nsample, points, idx, out, 1);
}

void eda_oracle_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                  const float *grad_out, const int *idx,
                                  float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);   /* group_points.cpp:52-54 */
  for (int i = 0; i < b; i++)
    for (int l = 0; l < c; l++) {
      float *gp = grad_points + ((size_t)i * c + l) * n;
      const int *ix = idx + (size_t)i * npoints * nsample;
      const float *go = grad_out + ((size_t)i * c + l) * npoints * nsample;
      for (int j = 0; j < npoints; j++)
        for (int k = 0; k < nsample; ++k)   /* group_points_gpu.cu:63-67; ascending (j,k) */
          gp[ix[(size_t)j * nsample + k]] += go[(size_t)j * nsample + k];
    }
}

/* ---------------------------------------------------------- interpolate */

EDA_CLONES
void eda_oracle_three_nn(int b, int n, int m, const float *unknown,
                         const float *known, float *dist2, int *idx) {
  const int mode = g_fma_mode;
  for (int i = 0; i < b; i++) {
    const float *un = unknown + (size_t)i * n * 3;
    const float *kn = known + (size_t)i * m * 3;
    float *d2o = dist2 + (size_t)i * n * 3;
    int *io = idx + (size_t)i * n * 3;
    for (int j = 0; j < n; j++) {
      const float ux = un[j * 3 + 0], uy = un[j * 3 + 1], uz = un[j * 3 + 2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;     /* interpolate_gpu.cu:33 */
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float x = kn[k * 3 + 0], y = kn[k * 3 + 1], z = kn[k * 3 + 2];
        const float d = sumsq3(ux - x, uy - y, uz - z, mode);   /* :39 */
        if (d < best1) {                                         /* :40-56 */
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      d2o[j * 3 + 0] = (float)best1;     /* :58-60 double -> float store */
      d2o[j * 3 + 1] = (float)best2;
      d2o[j * 3 + 2] = (float)best3;
      io[j * 3 + 0] = besti1;
      io[j * 3 + 1] = besti2;
      io[j * 3 + 2] = besti3;
    }
  }
}

/* p1*w1 + p2*w2 + p3*w3 (interpolate_gpu.cu:103-104) under the same
 * contraction rule: t = p2*w2; t = fma(p1,w1,t); t = fma(p3,w3,t).        */
static inline float interp3(float p1, float w1, float p2, float w2, float p3,
                            float w3, int mode) {
  if (mode == 0) {
    float t = p2 * w2;
    t = fmaf(p1, w1, t);
    t = fmaf(p3, w3, t);
    return t;
  }
  return (p1 * w1 + p2 * w2) + p3 * w3;
}

void eda_oracle_three_interpolate(int b, int c, int m, int n,
                                  const float *points, const int *idx,
                                  const float *weight, float *out) {
  const int mode = g_fma_mode;
  for (int i = 0; i < b; i++)
    for (int l = 0; l < c; l++) {
      const float *p = points + ((size_t)i * c + l) * m;
      float *o = out + ((size_t)i * c + l) * n;
      for (int j = 0; j < n; j++) {
        const float *w = weight + ((size_t)i * n + j) * 3;
        const int *ix = idx + ((size_t)i * n + j) * 3;
        o[j] = interp3(p[ix[0]], w[0], p[ix[1]], w[1], p[ix[2]], w[2], mode);
      }
    }
}

void eda_oracle_three_interpolate_grad(int b, int c, int n, int m,
                                       const float *grad_out, const int *idx,
                                       const float *weight, float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * m);   /* interpolate.cpp:90-92 */
  for (int i = 0; i < b; i++)
    for (int l = 0; l < c; l++) {
      float *gp = grad_points + ((size_t)i * c + l) * m;
      const float *go = grad_out + ((size_t)i * c + l) * n;
      for (int j = 0; j < n; j++) {          /* interpolate_gpu.cu:133-147, ascending j */
        const float *w = weight + ((size_t)i * n + j) * 3;
        const int *ix = idx + ((size_t)i * n + j) * 3;
        gp[ix[0]] += go[j] * w[0];
        gp[ix[1]] += go[j] * w[1];
        gp[ix[2]] += go[j] * w[2];
      }
    }
}
