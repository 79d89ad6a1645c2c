"""Global-batch BatchNorm statistics INSIDE the fused set-abstraction / feature-propagation calls (the reference's
SyncBatchNorm, main_utils.py:336-338; include/eda_hip.h: eda_set_bn_sync; eda_amd/sync_bn.py).  No second GPU is needed
to check the mechanism:
  * world = 1 with a hook that reduces nothing: the split path (statistics left un-finalised by the GEMM epilogue, hook,
    separate finalise kernel) must equal the fused call BIT FOR BIT, forward and backward;
  * world = 2 emulated with a hook that doubles the sums (two ranks holding the same rows): global statistics equal
    the local ones, so outputs and gradients must again equal the plain call's (d(gamma), d(beta) = global sums / world);
  * the hook is called once per layer and direction, with 2 C doubles."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(gather, seed=0):
    from eda_amd import sa_ops
    torch.manual_seed(seed)
    dev = "cuda"
    chans = [16, 32, 64, 64] if not gather else [3 + 8, 32, 64]
    L = len(chans) - 1
    params = []
    for l in range(L):
        w = (torch.randn(chans[l + 1], chans[l], device=dev) / chans[l] ** 0.5).requires_grad_(True)
        g = (torch.rand(chans[l + 1], device=dev) + 0.5).requires_grad_(True)
        b = (torch.randn(chans[l + 1], device=dev) * 0.1).requires_grad_(True)
        params += [w, g, b]
    running = [(torch.zeros(chans[l + 1], device=dev), torch.ones(chans[l + 1], device=dev)) for l in range(L)]
    if gather:
        B, N, m, ns = 2, 512, 64, 16
        xyz = torch.rand(B, N, 3, device=dev)
        new_xyz = xyz[:, :m].contiguous()
        feats = torch.randn(B, N, 8, device=dev, requires_grad=True)
        idx = torch.randint(0, N, (B, m, ns), device=dev, dtype=torch.int32)
        cfg = dict(gather=True, radius=0.5, normalize_xyz=True, pool=ns, training=True, eps=1e-5, momentum=0.1, running=running)
        out = sa_ops.FusedMLP.apply(cfg, None, xyz, new_xyz, feats, idx, *params)
        leaf = feats
    else:
        x = torch.randn(4096, chans[0], device=dev, requires_grad=True)
        cfg = dict(gather=False, radius=1.0, normalize_xyz=False, pool=16, training=True, eps=1e-5, momentum=0.1, running=running)
        out = sa_ops.FusedMLP.apply(cfg, x, None, None, None, None, *params)
        leaf = x
    w = torch.randn_like(out)
    (out * w).sum().backward()
    torch.cuda.synchronize()
    return [out.detach()] + [leaf.grad] + [p.grad for p in params] + [r for pair in running for r in pair]


@pytest.mark.parametrize("gather", [False, True])
def test_split_statistics_path_equals_the_fused_call(gather):
    from eda_amd import sync_bn
    plain = _run(gather)
    calls = []
    sync_bn._reduce = lambda v: calls.append(v.numel())
    sync_bn.install_fused_hook(1)
    try:
        split = _run(gather)
    finally:
        sync_bn.remove_fused_hook()
        sync_bn._reduce = None
    L = 3 if not gather else 2
    assert len(calls) == 2 * L and all(n % 2 == 0 for n in calls), calls       # one per layer and direction
    for a, b in zip(plain, split):
        if gather and a.shape == b.shape and a.dim() == 3:
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)             # (scatter-add gradient: fp32 atomics order)
        else:
            assert torch.equal(a, b)


@pytest.mark.parametrize("gather", [False, True])
def test_two_ranks_with_identical_rows_emulated(gather):
    from eda_amd import sync_bn
    plain = _run(gather)
    sync_bn._reduce = lambda v: v.mul_(2.0)          # what all_reduce gives when the other rank holds the same rows
    sync_bn.install_fused_hook(2)
    try:
        two = _run(gather)
    finally:
        sync_bn.remove_fused_hook()
        sync_bn._reduce = None
    n_run = len(plain) - 2 * (3 if not gather else 2)
    for i, (a, b) in enumerate(zip(plain, two)):
        if i < n_run:
            # outputs, input gradient, dW, d(gamma), d(beta): global statistics = local ones
            torch.testing.assert_close(a, b, rtol=2e-6, atol=1e-6)
    # running variance: unbiased with the GLOBAL count 2R instead of R
    rows = 4096 if not gather else 2 * 64 * 16
    rv_plain, rv_two = plain[-1], two[-1]
    expect = 0.9 + (rv_plain - 0.9) * ((rows - 1) / rows) * (2 * rows / (2 * rows - 1))
    torch.testing.assert_close(rv_two, expect, rtol=1e-5, atol=1e-6)


def test_native_exchange_on_one_rank_equals_the_plain_kernels_bit_for_bit():
    """`sync_bn.enable(native=True)` in a ONE-rank group: every BatchNorm kernel runs its peer exchange (csrc/peer.h) against
    its own slab -- the N > 1 code path on one GPU.  The sums are the local ones, so the fused SA / FP call, the single-launch
    BN+ReLU(+Dropout) kernels (R <= 4096), the multi-launch form (R > 4096, pooled too) and the stand-alone vector exchange
    must reproduce the plain results exactly, forward and backward; no exchange may time out."""
    import os
    import socket
    import torch.distributed as dist
    from eda_amd import _lib, sa_ops, sync_bn
    dev = "cuda"

    def small(R, C, pool=1, p=0.0):
        torch.manual_seed(R + C)
        z = torch.randn(R, C, device=dev, requires_grad=True)
        g = (torch.rand(C, device=dev) + 0.5).requires_grad_(True)
        b = (torch.randn(C, device=dev) * 0.1).requires_grad_(True)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        out = sa_ops.BNReLUCL.apply(z, g, b, rm, rv, 1e-5, 0.1, True, pool, p, 77)
        (out * torch.linspace(0.5, 1.5, out.numel(), device=dev).view_as(out)).sum().backward()
        return [out.detach(), z.grad, g.grad, b.grad, rm, rv]

    def everything():
        return _run(False) + _run(True) + small(2048, 288, p=0.1) + small(300, 64) + small(8192, 288) + small(8192, 128, pool=16)

    plain = everything()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        sync_bn.enable(single_rank_too=True, native=True)
        assert sync_bn.native() and not sync_bn.diverts() and _lib.lib().eda_peer_connected() == 1
        native = everything()
        buf = torch.arange(601, dtype=torch.float64, device=dev) * 0.25
        _lib.check(_lib.lib().eda_peer_allreduce_f64(buf.data_ptr(), buf.numel(), torch.cuda.current_stream().cuda_stream), "peer")
        assert torch.equal(buf, torch.arange(601, dtype=torch.float64, device=dev) * 0.25)
        assert sync_bn.peer_timeouts() == 0
    finally:
        sync_bn.disable()
        dist.destroy_process_group()
    for i, (a, b) in enumerate(zip(plain, native)):
        if a.dim() == 3:                                  # (scatter-add gradient of the gathered stack: fp32 atomics order)
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(a, b), i


def test_peer_self_test_passes_detects_an_injected_wrong_tag_and_recovers():
    """csrc/peer.hip eda_peer_selftest, one rank against its own slab: (1) the slab is fine-grained or
    uncached device memory and exports an IPC handle; (2) the exchange of the known vector passes; (3) with a wrong tag
    injected the polls run into their bound, the call returns EDA_ERR_PEER_SELFTEST, `sync_bn.check()`-style reading of the
    timeout word is non-zero until the reset; (4) after the reset it passes again; (5) `enable(native=True)` with the
    injected fault logs and falls back to the collective hook instead of switching the kernels to the peer exchange."""
    import os
    import socket
    import torch.distributed as dist
    from eda_amd import _lib, sync_bn
    L = _lib.lib()
    old = os.environ.get("EDA_PEER_SPIN_LOG2")
    os.environ["EDA_PEER_SPIN_LOG2"] = "10"
    L.eda_reload_env()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        sync_bn.enable(single_rank_too=True, native=True)
        assert sync_bn.native()
        assert L.eda_peer_alloc_kind() in (0, 1), L.eda_peer_alloc_kind()
        stream = torch.cuda.current_stream().cuda_stream
        assert L.eda_peer_selftest(stream, 0) == 0
        assert L.eda_peer_selftest(stream, 1) == 10004 and b"timed-out" in L.eda_last_error_string()
        assert sync_bn.peer_timeouts() > 0
        try:
            sync_bn.check()
            raise AssertionError("sync_bn.check() must raise on a counted timeout")
        except RuntimeError as e:
            assert "timed out" in str(e)
        _lib.check(L.eda_peer_reset(), "reset")
        assert sync_bn.peer_timeouts() == 0 and L.eda_peer_selftest(stream, 0) == 0
        sync_bn.disable()
        assert L.eda_peer_connected() == 0
        # the same fault at enable(): no native exchange, the collective hook instead
        sync_bn._selftest_inject = True
        sync_bn.enable(single_rank_too=True, native=True)
        assert not sync_bn.native() and sync_bn.fused_hook_installed() and sync_bn.diverts()
        assert sync_bn.peer_timeouts() == 0
    finally:
        sync_bn._selftest_inject = False
        sync_bn.disable()
        dist.destroy_process_group()
        if old is None:
            os.environ.pop("EDA_PEER_SPIN_LOG2", None)
        else:
            os.environ["EDA_PEER_SPIN_LOG2"] = old
        L.eda_reload_env()


def test_grouped_batchnorm_rejects_more_channels_than_the_peer_slab_has_granules():
    """granule index = matrix * C + channel must stay below PEER_MAXG (8192) when the kernels exchange their
    statistics through the slab -- 8 matrices x 4 groups x 288 channels = 9216 would write past one parity's region."""
    import ctypes
    import os
    import socket
    import torch.distributed as dist
    from eda_amd import _lib, sync_bn
    L = _lib.lib()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        sync_bn.enable(single_rank_too=True, native=True)
        nmat, ng, cpg, R = 8, 4, 288, 64
        C = ng * cpg
        dev = "cuda"
        z = [torch.randn(R, C, device=dev) for _ in range(nmat)]
        dz = [torch.empty(R, C, device=dev) for _ in range(nmat)]
        gam = [torch.ones(cpg, device=dev) for _ in range(nmat * ng)]
        st = [torch.zeros(4 * C, device=dev) for _ in range(nmat)]
        dgb = [torch.zeros(2 * C, device=dev) for _ in range(nmat)]
        arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        rc = L.eda_bn_relu_grouped_bwd_multi_f32(nmat, arr(z), arr(z), R, ng, cpg, arr(gam), arr(st), 1, arr(dgb), arr(dz), 0.0,
                                                 None, None, torch.cuda.current_stream().cuda_stream)
        assert rc == 10001 and b"PEER_MAXG" in L.eda_last_error_string()
    finally:
        sync_bn.disable()
        dist.destroy_process_group()
