// ball_query.hip -- radius neighbour query for gfx950.
//
// Replaces query_ball_point_kernel (reference
// pointnet2/_ext_src/src/ball_query_gpu.cu:14-49; host ball_query.cpp:13-37).
// Result contract (bit-exact): for centre j the first `nsample` indices k, in
// ascending k, with d2(new_xyz[j], xyz[k]) < radius^2 (fp32, strict); the row
// is padded with the first hit; a centre with no hit keeps an all-zero row.
//
// Machine mapping.  The reference uses one THREAD per centre and one block per
// scene (8 of 256 CUs busy at B = 8).  Here one WAVE owns CPW centres and the
// 64 lanes test 64 consecutive points at once: hits are appended in index order
// with ballot + prefix-popcount, so "first nsample by ascending index" holds by
// construction.  A workgroup (4 waves, 16 centres) stages each 1024-point chunk
// of the scene once in LDS (AoS; a stride-3 dword read is bank-conflict free on
// the 32-bank ds_read_b32 path) so HBM/L2 traffic per scene is n*12 bytes per
// 16 centres instead of per centre.  Grid = b * ceil(m / 16) workgroups.
#include "eda_common.h"

namespace {

constexpr int BQ_THREADS = 256;
constexpr int BQ_WAVES = BQ_THREADS / 64;
constexpr int BQ_CPW = 4;                       // centres per wave
constexpr int BQ_CPB = BQ_WAVES * BQ_CPW;       // centres per workgroup
constexpr int BQ_CHUNK = 1024;                  // points staged per iteration

template <int MODE>
__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(
    const float *__restrict__ new_xyz_all, const float *__restrict__ xyz_all, int n, int m,
    float radius2, int nsample, int *__restrict__ idx_all, int blocks_per_scene) {
  __shared__ float pts[BQ_CHUNK * 3];

  const int scene = blockIdx.x / blocks_per_scene;
  const int cblk = blockIdx.x % blocks_per_scene;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  const float *xyz = xyz_all + (size_t)scene * n * 3;
  const float *new_xyz = new_xyz_all + (size_t)scene * m * 3;
  int *idx = idx_all + (size_t)scene * m * nsample;

  const int j0 = cblk * BQ_CPB + wave * BQ_CPW;   // first centre of this wave

  float cx[BQ_CPW], cy[BQ_CPW], cz[BQ_CPW];
  int cnt[BQ_CPW], first[BQ_CPW];
#pragma unroll
  for (int c = 0; c < BQ_CPW; ++c) {
    const int j = j0 + c;
    const bool live = j < m;
    cx[c] = live ? new_xyz[j * 3 + 0] : 0.f;
    cy[c] = live ? new_xyz[j * 3 + 1] : 0.f;
    cz[c] = live ? new_xyz[j * 3 + 2] : 0.f;
    cnt[c] = live ? 0 : nsample;                  // dead centres are "done"
    first[c] = 0;
  }
  const unsigned long long lt_mask = (1ull << lane) - 1ull;

  for (int base = 0; base < n; base += BQ_CHUNK) {
    // Workgroup-uniform early exit: every centre of every wave already full.
    bool wave_done = true;
#pragma unroll
    for (int c = 0; c < BQ_CPW; ++c) wave_done = wave_done && (cnt[c] >= nsample);
    if (__syncthreads_and(wave_done)) break;      // also orders LDS reuse

    const int npts = min(BQ_CHUNK, n - base);
    const float *src = xyz + (size_t)base * 3;
    for (int f = tid; f < npts * 3; f += BQ_THREADS) pts[f] = src[f];   // coalesced dwords
    __syncthreads();

    if (!wave_done) {
      for (int g = 0; g < npts; g += 64) {
        const int p = g + lane;
        const bool in = p < npts;
        const float x = in ? pts[p * 3 + 0] : 0.f;
        const float y = in ? pts[p * 3 + 1] : 0.f;
        const float z = in ? pts[p * 3 + 2] : 0.f;
        const int k = base + p;
#pragma unroll
        for (int c = 0; c < BQ_CPW; ++c) {
          if (cnt[c] >= nsample) continue;         // wave-uniform
          // centre minus point (ball_query_gpu.cu:36-37)
          const float d2 = eda_sumsq3<MODE>(cx[c] - x, cy[c] - y, cz[c] - z);
          const bool hit = in && (d2 < radius2);
          const unsigned long long mask = __ballot(hit);
          if (mask == 0ull) continue;
          if (cnt[c] == 0) first[c] = base + g + (__ffsll((long long)mask) - 1);
          const int rank = cnt[c] + __popcll(mask & lt_mask);
          if (hit && rank < nsample) idx[(size_t)(j0 + c) * nsample + rank] = k;
          cnt[c] += __popcll(mask);
        }
      }
    }
  }

  // pad with the first hit (ball_query_gpu.cu:39-43); empty ball -> zeros
#pragma unroll
  for (int c = 0; c < BQ_CPW; ++c) {
    const int j = j0 + c;
    if (j >= m) continue;
    const int have = min(cnt[c], nsample);
    for (int s = have + lane; s < nsample; s += 64) idx[(size_t)j * nsample + s] = first[c];
  }
}


// =====================================================================================
// Grid-accelerated variant (large scenes).  The brute-force scan above tests every point
// against every centre (1.0e8 tests per 50 000-point scene).  Here the points of a scene
// are binned into a uniform grid whose cells are at least 1.001 * radius wide (<= 32^3
// cells), so every point within the radius of a centre lies in the 3x3x3 cell
// neighbourhood of the centre's cell.  The three x-neighbours are adjacent in the sorted
// array, so a centre scans 9 contiguous candidate ranges.  Hits come out in cell order, not
// index order; the reference's "first nsample by ascending index" is restored exactly by
// ranking the hits of a centre by index (they are distinct) and keeping ranks < nsample.
// Distance arithmetic is the same eda_sumsq3 as everywhere, so the result is bit-identical
// to the scan.  A centre with more than GQ_CAP hits (never seen on room scenes) falls back
// to the index-ordered scan inside the same kernel.
constexpr int GQ_MAXG = 32;                        // cells per axis
constexpr int GQ_NC = GQ_MAXG * GQ_MAXG * GQ_MAXG;  // cell table size per scene
constexpr int GQ_CAP = 1024;                       // hit list capacity per centre
constexpr int GQ_THREADS = 256;

struct GridParams {          // per scene, written by the bbox kernel
  float minx, miny, minz;
  float invx, invy, invz;    // cells per metre
  int gx, gy, gz;
  int pad;
};

__device__ __forceinline__ int cell_coord(float v, float mn, float inv, int g) {
  int c = (int)floorf((v - mn) * inv);
  return c < 0 ? 0 : (c >= g ? g - 1 : c);
}

// one workgroup per scene: bounding box -> grid geometry
__global__ __launch_bounds__(1024) void gq_bbox_kernel(const float *__restrict__ xyz_all, int n,
                                                       float radius, GridParams *__restrict__ gp) {
  __shared__ float red[6][16];
  const float *xyz = xyz_all + (size_t)blockIdx.x * n * 3;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < n; i += 1024)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float v = xyz[(size_t)i * 3 + d];
      mn[d] = fminf(mn[d], v);
      mx[d] = fmaxf(mx[d], v);
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    for (int o = 32; o > 0; o >>= 1) {
      mn[d] = fminf(mn[d], __shfl_xor(mn[d], o));
      mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o));
    }
    if (lane == 0) { red[d][wave] = mn[d]; red[3 + d][wave] = mx[d]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    GridParams g;
    float lo[3], ext[3];
    int G[3];
    float inv[3];
    for (int d = 0; d < 3; ++d) {
      float a = INFINITY, b = -INFINITY;
      for (int w = 0; w < 16; ++w) { a = fminf(a, red[d][w]); b = fmaxf(b, red[3 + d][w]); }
      lo[d] = a;
      ext[d] = b - a;
      const float cell = radius * 1.001f;
      int cnt = (ext[d] > 0.f && cell > 0.f) ? (int)floorf(ext[d] / cell) : 1;
      if (!(cnt >= 1)) cnt = 1;                 // also catches NaN
      if (cnt > GQ_MAXG) cnt = GQ_MAXG;
      G[d] = cnt;
      inv[d] = ext[d] > 0.f ? (float)cnt / ext[d] : 0.f;   // cell width ext/cnt >= 1.001 * radius
    }
    g.minx = lo[0]; g.miny = lo[1]; g.minz = lo[2];
    g.invx = inv[0]; g.invy = inv[1]; g.invz = inv[2];
    g.gx = G[0]; g.gy = G[1]; g.gz = G[2]; g.pad = 0;
    gp[blockIdx.x] = g;
  }
}

// cell id of every point + histogram; the atomic's return value is the slot inside the cell
__global__ __launch_bounds__(256) void gq_count_kernel(const float *__restrict__ xyz_all, int n,
                                                       const GridParams *__restrict__ gp,
                                                       int *__restrict__ cell_count,
                                                       int *__restrict__ pt_cell,
                                                       int *__restrict__ pt_slot) {
  const int scene = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const GridParams g = gp[scene];
  const float *p = xyz_all + ((size_t)scene * n + i) * 3;
  const int cx = cell_coord(p[0], g.minx, g.invx, g.gx);
  const int cy = cell_coord(p[1], g.miny, g.invy, g.gy);
  const int cz = cell_coord(p[2], g.minz, g.invz, g.gz);
  const int cid = (cz * g.gy + cy) * g.gx + cx;
  pt_cell[(size_t)scene * n + i] = cid;
  pt_slot[(size_t)scene * n + i] = atomicAdd(cell_count + (size_t)scene * GQ_NC + cid, 1);
}

// exclusive scan of the (<= 32768) cell counts of a scene; one workgroup per scene
__global__ __launch_bounds__(1024) void gq_scan_kernel(const int *__restrict__ cell_count,
                                                       int *__restrict__ cell_start) {
  __shared__ int part[1024];
  const int *cnt = cell_count + (size_t)blockIdx.x * GQ_NC;
  int *st = cell_start + (size_t)blockIdx.x * (GQ_NC + 1);
  constexpr int PER = GQ_NC / 1024;       // 32 consecutive cells per thread
  int local[PER];
  int s = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) { local[k] = s; s += cnt[threadIdx.x * PER + k]; }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {      // Hillis-Steele inclusive scan of the per-thread sums
    const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  const int base = threadIdx.x ? part[threadIdx.x - 1] : 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) st[threadIdx.x * PER + k] = base + local[k];
  if (threadIdx.x == 1023) st[GQ_NC] = part[1023];
}

// sorted[cell_start[cell] + slot] = (x, y, z, original index)
__global__ __launch_bounds__(256) void gq_scatter_kernel(const float *__restrict__ xyz_all, int n,
                                                         const int *__restrict__ cell_start,
                                                         const int *__restrict__ pt_cell,
                                                         const int *__restrict__ pt_slot,
                                                         float4 *__restrict__ sorted) {
  const int scene = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float *p = xyz_all + ((size_t)scene * n + i) * 3;
  const int cid = pt_cell[(size_t)scene * n + i];
  const int dst = cell_start[(size_t)scene * (GQ_NC + 1) + cid] + pt_slot[(size_t)scene * n + i];
  sorted[(size_t)scene * n + dst] = make_float4(p[0], p[1], p[2], __int_as_float(i));
}

template <int MODE>
__global__ __launch_bounds__(GQ_THREADS) void gq_query_kernel(
    const float *__restrict__ new_xyz_all, const float *__restrict__ xyz_all, int n, int m,
    float radius2, int nsample, const GridParams *__restrict__ gp,
    const int *__restrict__ cell_start_all, const float4 *__restrict__ sorted_all,
    int *__restrict__ idx_all, int blocks_per_scene) {
  __shared__ int hits[GQ_THREADS / 64][GQ_CAP];
  const int scene = blockIdx.x / blocks_per_scene;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = (blockIdx.x % blocks_per_scene) * (GQ_THREADS / 64) + wave;
  if (j >= m) return;                               // whole wave exits (no block barriers below)
  const GridParams g = gp[scene];
  const float *ctr = new_xyz_all + ((size_t)scene * m + j) * 3;
  const float cx = ctr[0], cy = ctr[1], cz = ctr[2];
  const int *cell_start = cell_start_all + (size_t)scene * (GQ_NC + 1);
  const float4 *sorted = sorted_all + (size_t)scene * n;
  int *row = idx_all + ((size_t)scene * m + j) * nsample;
  int *list = hits[wave];
  const unsigned long long lt_mask = (1ull << lane) - 1ull;

  // unclamped cell of the centre, limited to [-1, G] so that a far-away centre scans nothing extra
  auto ccell = [](float v, float mn, float inv, int gdim) {
    const float f = floorf((v - mn) * inv);
    return f < -1.f ? -1 : (f > (float)gdim ? gdim : (int)f);
  };
  const int ix = ccell(cx, g.minx, g.invx, g.gx), iy = ccell(cy, g.miny, g.invy, g.gy),
            iz = ccell(cz, g.minz, g.invz, g.gz);
  const int x0 = max(ix - 1, 0), x1 = min(ix + 1, g.gx - 1);
  int H = 0;
  if (x0 <= x1) {
    for (int zz = max(iz - 1, 0); zz <= min(iz + 1, g.gz - 1); ++zz)
      for (int yy = max(iy - 1, 0); yy <= min(iy + 1, g.gy - 1); ++yy) {
        const int rowbase = (zz * g.gy + yy) * g.gx;
        const int beg = cell_start[rowbase + x0], end = cell_start[rowbase + x1 + 1];
        for (int i0 = beg; i0 < end; i0 += 64) {
          const int i = i0 + lane;
          bool hit = false;
          int pidx = 0;
          if (i < end) {
            const float4 p = sorted[i];
            const float d2 = eda_sumsq3<MODE>(cx - p.x, cy - p.y, cz - p.z);   // centre minus point
            hit = d2 < radius2;
            pidx = __float_as_int(p.w);
          }
          const unsigned long long mask = __ballot(hit);
          if (mask == 0ull) continue;
          const int pos = H + __popcll(mask & lt_mask);
          if (hit && pos < GQ_CAP) list[pos] = pidx;
          H += __popcll(mask);
        }
      }
  }

  if (H > GQ_CAP) {
    // overflow (pathologically dense ball): index-ordered scan over the whole scene
    const float *xyz = xyz_all + (size_t)scene * n * 3;
    int cnt = 0, first = 0;
    for (int k0 = 0; k0 < n && cnt < nsample; k0 += 64) {
      const int k = k0 + lane;
      bool hit = false;
      if (k < n) hit = eda_sumsq3<MODE>(cx - xyz[(size_t)k * 3], cy - xyz[(size_t)k * 3 + 1],
                                        cz - xyz[(size_t)k * 3 + 2]) < radius2;
      const unsigned long long mask = __ballot(hit);
      if (mask == 0ull) continue;
      if (cnt == 0) first = k0 + (__ffsll((long long)mask) - 1);
      const int rank = cnt + __popcll(mask & lt_mask);
      if (hit && rank < nsample) row[rank] = k;
      cnt += __popcll(mask);
    }
    for (int sidx = min(cnt, nsample) + lane; sidx < nsample; sidx += 64) row[sidx] = first;
    return;
  }

  // rank every hit by its point index (indices are distinct); ranks < nsample are the answer
  int first = 0;
  for (int base = 0; base < H; base += 64) {
    const int e = base + lane;
    const int v = e < H ? list[e] : 0x7fffffff;
    int rank = 0;
    for (int t = 0; t < H; ++t) rank += (list[t] < v) ? 1 : 0;     // LDS broadcast reads
    if (e < H && rank < nsample) row[rank] = v;
    const unsigned long long zero = __ballot(e < H && rank == 0);
    if (zero) first = __shfl(v, __ffsll((long long)zero) - 1);
  }
  for (int sidx = min(H, nsample) + lane; sidx < nsample; sidx += 64) row[sidx] = first;   // H == 0 -> zeros
}

}  // namespace

extern "C" size_t eda_ball_query_workspace_bytes(int b, int n, int m) {
  (void)m;
  if (b <= 0 || n <= 0) return 0;
  size_t per = sizeof(GridParams) + sizeof(int) * (size_t)GQ_NC + sizeof(int) * (size_t)(GQ_NC + 1) +
               sizeof(int) * 2 * (size_t)n + sizeof(float4) * (size_t)n + 64;
  return (size_t)b * per + 1024;     // + alignment slack for the carve-up below
}

static int launch_scan(const float *new_xyz, const float *xyz, int b, int n, int m, float radius2,
                       int nsample, int *idx, hipStream_t stream) {
  const int bps = (m + BQ_CPB - 1) / BQ_CPB;
  const dim3 grid((unsigned)((size_t)b * bps));
  if (g_eda_fma_mode == 0)
    hipLaunchKernelGGL(ball_query_kernel<0>, grid, dim3(BQ_THREADS), 0, stream, new_xyz, xyz, n, m,
                       radius2, nsample, idx, bps);
  else
    hipLaunchKernelGGL(ball_query_kernel<1>, grid, dim3(BQ_THREADS), 0, stream, new_xyz, xyz, n, m,
                       radius2, nsample, idx, bps);
  EDA_CHECK_LAUNCH();
  return 0;
}

// ws may be NULL (or too small): then the index-ordered scan kernel is used for any n.
extern "C" int eda_ball_query_f32(const float *new_xyz, const float *xyz, int b, int n, int m,
                                  float radius, int nsample, int *idx, void *ws, size_t ws_bytes,
                                  void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, "negative dimension");
  if (b == 0 || m == 0 || nsample == 0) return 0;
  EDA_CHECK_ARG(new_xyz && idx && (xyz || n == 0), "null pointer");
  const float radius2 = radius * radius;          // ball_query_gpu.cu:26, fp32
  const bool use_grid = ws && n >= 4096 && b <= 65535 && radius > 0.f &&
                        ws_bytes >= eda_ball_query_workspace_bytes(b, n, m) && !eda_knob_set(EDA_K_BQ_SCAN);
  if (!use_grid) return launch_scan(new_xyz, xyz, b, n, m, radius2, nsample, idx, stream);

  // carve the workspace (all offsets 16-byte aligned)
  unsigned char *w = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
  GridParams *gp = reinterpret_cast<GridParams *>(w);            w += ((sizeof(GridParams) * b + 63) / 64) * 64;
  int *cell_count = reinterpret_cast<int *>(w);                  w += sizeof(int) * (size_t)b * GQ_NC;
  int *cell_start = reinterpret_cast<int *>(w);                  w += ((sizeof(int) * (size_t)b * (GQ_NC + 1) + 63) / 64) * 64;
  int *pt_cell = reinterpret_cast<int *>(w);                     w += ((sizeof(int) * (size_t)b * n + 63) / 64) * 64;
  int *pt_slot = reinterpret_cast<int *>(w);                     w += ((sizeof(int) * (size_t)b * n + 63) / 64) * 64;
  float4 *sorted = reinterpret_cast<float4 *>(w);

  { const int zrc__ = eda_zero_async(cell_count, sizeof(int) * (size_t)b * GQ_NC, stream); if (zrc__) return zrc__; }
  hipLaunchKernelGGL(gq_bbox_kernel, dim3(b), dim3(1024), 0, stream, xyz, n, radius, gp);
  EDA_CHECK_LAUNCH();
  const dim3 pgrid((unsigned)((n + 255) / 256), (unsigned)b);
  hipLaunchKernelGGL(gq_count_kernel, pgrid, dim3(256), 0, stream, xyz, n, gp, cell_count, pt_cell, pt_slot);
  EDA_CHECK_LAUNCH();
  hipLaunchKernelGGL(gq_scan_kernel, dim3(b), dim3(1024), 0, stream, cell_count, cell_start);
  EDA_CHECK_LAUNCH();
  hipLaunchKernelGGL(gq_scatter_kernel, pgrid, dim3(256), 0, stream, xyz, n, cell_start, pt_cell, pt_slot, sorted);
  EDA_CHECK_LAUNCH();
  const int wpb = GQ_THREADS / 64;
  const int bps = (m + wpb - 1) / wpb;
  const dim3 qgrid((unsigned)((size_t)b * bps));
  if (g_eda_fma_mode == 0)
    hipLaunchKernelGGL(gq_query_kernel<0>, qgrid, dim3(GQ_THREADS), 0, stream, new_xyz, xyz, n, m, radius2,
                       nsample, gp, cell_start, sorted, idx, bps);
  else
    hipLaunchKernelGGL(gq_query_kernel<1>, qgrid, dim3(GQ_THREADS), 0, stream, new_xyz, xyz, n, m, radius2,
                       nsample, gp, cell_start, sorted, idx, bps);
  EDA_CHECK_LAUNCH();
  return 0;
}
