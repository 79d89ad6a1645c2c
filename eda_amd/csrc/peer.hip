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
int g_own_kind = -1;                      // how the slab was allocated: 0 fine-grained, 1 uncached, 2 plain (coarse-grained)
double *g_selftest_buf = nullptr;         // PEER_SELFTEST_N doubles (device) + the host copy the self-test compares
void *g_opened[PEER_MAXW] = {};
unsigned char g_opened_handle[PEER_MAXW][64] = {};
constexpr int PEER_SELFTEST_N = 64;

// The slab is polled and written by OTHER GPUs while this GPU's kernels run.  Plain hipMalloc memory is coarse-grained:
// coherent between agents at kernel boundaries only, so a local L2 could keep serving a stale tag line for the length of
// the polling kernel.  Fine-grained (or, failing that, uncached) device memory is what flag memory has to be; the
// allocation is tried in that order and must also export an IPC handle (EDA_PEER_ALLOC pins one kind: 0 / 1 / 2).
int alloc_slab(void **out, int *kind_out, hipIpcMemHandle_t *h) {
  const size_t bytes = PEER_SLAB_WORDS * sizeof(unsigned long long);
  const long pin = eda_knob_set(EDA_K_PEER_ALLOC) ? eda_knob(EDA_K_PEER_ALLOC) : -1;
  hipError_t last = hipErrorUnknown;
  for (int kind = 0; kind < 3; ++kind) {
    if (pin >= 0 && pin != kind) continue;
    void *p = nullptr;
    hipError_t e = kind == 0   ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained)
                   : kind == 1 ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached)
                               : hipMalloc(&p, bytes);
    if (e == hipSuccess) e = hipIpcGetMemHandle(h, p);
    if (e == hipSuccess) { *out = p; *kind_out = kind; return 0; }
    (void)hipGetLastError();
    if (p) (void)hipFree(p);
    last = e;
  }
  eda_set_error("eda_peer_create: no slab allocation exports an IPC handle: %s", hipGetErrorString(last));
  return (int)last;
}

// one exchange of a known vector: rank r contributes (r + 1) * (i + 1) to element i
__global__ __launch_bounds__(64) void peer_selftest_kernel(const EdaPeer P, double *__restrict__ buf, int wrong_tag) {
  unsigned long long seq = eda_peer_seq(P);
  const int g = threadIdx.x;
  if (g < PEER_SELFTEST_N / 2) {
    double a = (double)(P.rank + 1) * (2 * g + 1), b = (double)(P.rank + 1) * (2 * g + 2);
    // wrong_tag: publish under a sequence number nobody polls for (the injected fault of tests/test_sync_bn_gpu.py:
    // every poll runs into its bound, the timeout word counts it, eda_peer_selftest() reports the failure)
    eda_peer_publish2(P, wrong_tag ? seq + 2 : seq, g, a, b);
    eda_peer_poll2(P, seq, g, a, b);
    buf[2 * g] = a; buf[2 * g + 1] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) eda_peer_done(P, 1);
}

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
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  hipIpcMemHandle_t h;
  if (!g_own_slab) {
    int rc = alloc_slab(&g_own_slab, &g_own_kind, &h);
    if (rc) return rc;
    EDA_CHECK_HIP(hipMemset(g_own_slab, 0, eda_peer_slab_bytes()));
    EDA_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&g_selftest_buf), PEER_SELFTEST_N * sizeof(double)));
    EDA_CHECK_HIP(hipDeviceSynchronize());
  } else {
    EDA_CHECK_HIP(hipIpcGetMemHandle(&h, g_own_slab));
  }
  memcpy(handle_out, &h, 64);
  return 0;
}

// 0 fine-grained, 1 uncached, 2 plain coarse-grained device memory (only with EDA_PEER_ALLOC=2 or when nothing better exports
// an IPC handle -- safe between processes on ONE device, not across GPUs); -1 before eda_peer_create()
extern "C" int eda_peer_alloc_kind(void) { return g_own_kind; }

// Zero this rank's slab (sequence number, arrival ticket, timeout word, every granule and tag).  Every rank of the group calls
// it with no exchanging launch in flight anywhere, and a host barrier follows before the next one (eda_amd/sync_bn.py does
// both): a new session, or the recovery after a counted timeout.
extern "C" int eda_peer_reset(void) {
  EDA_CHECK_ARG(g_own_slab, "eda_peer_create() first");
  EDA_CHECK_HIP(hipDeviceSynchronize());
  EDA_CHECK_HIP(hipMemset(g_own_slab, 0, eda_peer_slab_bytes()));
  EDA_CHECK_HIP(hipDeviceSynchronize());
  return 0;
}

// handles: world x 64 bytes, rank order (this rank's own entry is ignored).  world == 1: no peers, the exchange runs
// against the own slab alone (the N > 1 code path on one GPU).  A peer whose handle differs from the one mapped before (a
// restarted process, another group or rank order) is re-opened.
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
  g_peer_on = false;
  for (int r = 0; r < PEER_MAXW; ++r) {
    const unsigned char *hr = r < world && r != rank ? reinterpret_cast<const unsigned char *>(handles) + 64 * (size_t)r : nullptr;
    if (g_opened[r] && (!hr || memcmp(g_opened_handle[r], hr, 64) != 0)) {
      (void)hipIpcCloseMemHandle(g_opened[r]);
      g_opened[r] = nullptr;
    }
    if (r >= world) continue;
    if (r == rank) { p.slab[r] = reinterpret_cast<unsigned long long *>(g_own_slab); continue; }
    if (!g_opened[r]) {
      hipIpcMemHandle_t h;
      memcpy(&h, hr, 64);
      EDA_CHECK_HIP(hipIpcOpenMemHandle(&g_opened[r], h, hipIpcMemLazyEnablePeerAccess));
      memcpy(g_opened_handle[r], hr, 64);
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

// One exchange of a known vector on `stream` (EVERY rank calls it, like any exchanging launch), then the host checks the
// sums and the timeout word: 0 = the slabs carry data between the ranks; EDA_ERR_PEER_SELFTEST = they do not (wrong sums or a
// poll that ran into its bound) -- the caller falls back to collectives (eda_amd/sync_bn.py).  Synchronises the stream.
// inject_wrong_tag != 0: this rank publishes under a sequence number nobody polls for (test seam).
extern "C" int eda_peer_selftest(void *stream_, int inject_wrong_tag) {
  EDA_CHECK_ARG(g_peer_on && g_selftest_buf, "eda_peer_connect() first");
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(peer_selftest_kernel, dim3(1), dim3(64), 0, stream, g_peer, g_selftest_buf, inject_wrong_tag);
  EDA_CHECK_LAUNCH();
  double host[PEER_SELFTEST_N];
  EDA_CHECK_HIP(hipMemcpyAsync(host, g_selftest_buf, sizeof(host), hipMemcpyDeviceToHost, stream));
  EDA_CHECK_HIP(hipStreamSynchronize(stream));
  const long to = eda_peer_timeouts();
  const double wsum = 0.5 * g_peer.world * (g_peer.world + 1);
  int bad = 0;
  for (int i = 0; i < PEER_SELFTEST_N; ++i) bad += host[i] != wsum * (i + 1);
  if (bad || to != 0) {
    eda_set_error("eda_peer_selftest: %d of %d sums wrong, %ld timed-out polls (rank %d of %d, slab kind %d)", bad,
                  PEER_SELFTEST_N, to, g_peer.rank, g_peer.world, g_own_kind);
    return EDA_ERR_PEER_SELFTEST;
  }
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
