// peer.hip -- setup of the peer-mapped slabs behind the in-kernel BatchNorm-statistics exchange (csrc/peer.h) and the
// stand-alone exchange of a device vector (the eda_bn_sync_fn hook of the fused set-abstraction calls, include/eda_hip.h).
//
// Replaces, for the reference's multi-GPU configuration (SyncBatchNorm, main_utils.py:336-338), the one RCCL all-reduce per
// BatchNorm layer and direction: no collective library call, nothing the host has to issue between two kernels of a layer,
// so the whole step stays ONE captured hipGraph.  One process per GPU; every process creates its slab, the IPC handles are
// exchanged once by the host (torch.distributed object collectives, eda_amd/sync_bn.py), every process maps the others'.
// Tested with two processes on ONE device (tests/test_two_rank_gpu.py); over xGMI the protocol is the same
// (system-scope write-through stores into the peers' memory, polls of local memory only) but UNMEASURED here.
#include "eda_common.h"
#include "peer.h"

#include <string.h>

namespace {

EdaPeer g_peer = {};
bool g_peer_on = false;
void *g_own_slab = nullptr;
void *g_opened[PEER_MAXW] = {};

// buf[0 .. n) <- sum over the ranks (n doubles = n / 2 granules of two, an odd tail rides with a zero partner)
__global__ __launch_bounds__(256) void peer_allreduce_kernel(const EdaPeer P, double *__restrict__ buf, long n) {
  const unsigned long long seq = eda_peer_seq(P);
  const long pairs = (n + 1) / 2;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < pairs; g += (long)gridDim.x * 256) {
    double a = buf[2 * g], b = 2 * g + 1 < n ? buf[2 * g + 1] : 0.0;
    eda_peer_exchange2(P, seq, (int)g, a, b);
    buf[2 * g] = a;
    if (2 * g + 1 < n) buf[2 * g + 1] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) eda_peer_done(P, gridDim.x);
}

}  // namespace

const EdaPeer *eda_peer_active() { return g_peer_on ? &g_peer : nullptr; }

extern "C" size_t eda_peer_slab_bytes(void) { return PEER_SLAB_WORDS * sizeof(unsigned long long); }

// Allocate and zero this process's slab; handle_out receives its 64-byte IPC handle.
extern "C" int eda_peer_create(void *handle_out) {
  EDA_CHECK_ARG(handle_out, "null pointer");
  if (!g_own_slab) {
    EDA_CHECK_HIP(hipMalloc(&g_own_slab, eda_peer_slab_bytes()));
    EDA_CHECK_HIP(hipMemset(g_own_slab, 0, eda_peer_slab_bytes()));
    EDA_CHECK_HIP(hipDeviceSynchronize());
  }
  hipIpcMemHandle_t h;
  EDA_CHECK_HIP(hipIpcGetMemHandle(&h, g_own_slab));
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  memcpy(handle_out, &h, 64);
  return 0;
}

// handles: world x 64 bytes, rank order (this rank's own entry is ignored).  world == 1: no peers, the exchange runs
// against the own slab alone (the N > 1 code path on one GPU).
extern "C" int eda_peer_connect(int rank, int world, const void *handles) {
  EDA_CHECK_ARG(world >= 1 && world <= PEER_MAXW && rank >= 0 && rank < world, "1..8 ranks");
  EDA_CHECK_ARG(g_own_slab, "eda_peer_create() first");
  EDA_CHECK_ARG(world == 1 || handles, "null pointer");
  EdaPeer p = {};
  p.rank = rank; p.world = world;
  {
    long lg = eda_knob(EDA_K_PEER_SPIN_LOG2);
    if (lg < 4) lg = 4;
    if (lg > 31) lg = 31;
    p.spin_limit = 1u << lg;
  }
  for (int r = 0; r < world; ++r) {
    if (r == rank) { p.slab[r] = reinterpret_cast<unsigned long long *>(g_own_slab); continue; }
    if (!g_opened[r]) {
      hipIpcMemHandle_t h;
      memcpy(&h, reinterpret_cast<const unsigned char *>(handles) + 64 * (size_t)r, 64);
      EDA_CHECK_HIP(hipIpcOpenMemHandle(&g_opened[r], h, hipIpcMemLazyEnablePeerAccess));
    }
    p.slab[r] = reinterpret_cast<unsigned long long *>(g_opened[r]);
  }
  g_peer = p;
  g_peer_on = true;
  return 0;
}

extern "C" int eda_peer_disconnect(void) {
  g_peer_on = false;
  for (int r = 0; r < PEER_MAXW; ++r)
    if (g_opened[r]) { (void)hipIpcCloseMemHandle(g_opened[r]); g_opened[r] = nullptr; }
  return 0;
}

extern "C" int eda_peer_connected(void) { return g_peer_on ? g_peer.world : 0; }

// sticky count of exchanges that gave up their bounded spin (host read: synchronises the device)
extern "C" long eda_peer_timeouts(void) {
  if (!g_own_slab) return 0;
  unsigned long long v = 0;
  if (hipMemcpy(&v, reinterpret_cast<unsigned long long *>(g_own_slab) + 2, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (long)v;
}

extern "C" int eda_peer_allreduce_f64(double *buf, long n, void *stream_) {
  EDA_CHECK_ARG(n >= 0 && n <= 2L * PEER_MAXG, "at most 16384 doubles per call");
  if (n == 0) return 0;
  EDA_CHECK_ARG(buf, "null pointer");
  EDA_CHECK_ARG(g_peer_on, "eda_peer_connect() first");
  const long pairs = (n + 1) / 2;
  hipLaunchKernelGGL(peer_allreduce_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, g_peer, buf, n);
  EDA_CHECK_LAUNCH();
  return 0;
}

// eda_bn_sync_fn with the native exchange behind it: register with eda_set_bn_sync(eda_peer_bn_hook, NULL, world)
extern "C" int eda_peer_bn_hook(void *user, double *buf, long n, void *stream) {
  (void)user;
  return eda_peer_allreduce_f64(buf, n, stream);
}
