"""Degenerate sizes and argument checks of the C entry points added for the training path
(column sums, weight gradients, fused LayerNorm / BatchNorm variants), through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _stream():
    return torch.cuda.current_stream().cuda_stream


def test_empty_inputs_produce_zero_gradients():
    from eda_amd import _lib
    from eda_amd.nn_utils import colsum, wgrad
    L = _lib.lib()
    out = torch.full((288,), 3.0, device="cuda")
    colsum(torch.empty(0, 288, device="cuda"), out=out)
    assert (out == 0).all()
    dW = torch.full((3, 288), 3.0, device="cuda")
    db = torch.full((3,), 3.0, device="cuda")
    cnt = torch.zeros(64, dtype=torch.int32, device="cuda")
    rc = L.eda_wcolsum_f32(None, 0, 288, 288, None, 3, 3, dW.data_ptr(), db.data_ptr(), None, 0, cnt.data_ptr(), _stream())
    assert rc == 0 and (dW == 0).all() and (db == 0).all()
    dW2 = torch.full((288, 288), 3.0, device="cuda")
    db2 = torch.full((288,), 3.0, device="cuda")
    rc = L.eda_wgrad_f32(None, 288, None, 288, 0, 288, 288, dW2.data_ptr(), db2.data_ptr(), None, 0, _stream())
    assert rc == 0 and (dW2 == 0).all() and (db2 == 0).all()
    assert L.eda_wgrad_grouped_f32(None, 0, None, None, _stream()) == 0
    assert L.eda_ln_reduce_grouped_f32(None, 0, 288, _stream()) == 0
    # python-level: zero-row linear goes through the stock path and still yields zero gradients
    dWz, dbz = wgrad(torch.empty(0, 288, device="cuda"), torch.empty(0, 288, device="cuda"))
    assert dWz.abs().sum().item() == 0 and dbz.abs().sum().item() == 0


def test_argument_checks_report_errors():
    from eda_amd import _lib
    L = _lib.lib()
    x = torch.zeros(64, 290, device="cuda")
    out = torch.zeros(4, 290, device="cuda")
    cnt = torch.zeros(64, dtype=torch.int32, device="cuda")
    # C not a multiple of 4 is refused by the weighted column sums
    rc = L.eda_wcolsum_f32(x.data_ptr(), 64, 290, 290, x.data_ptr(), 290, 2, out.data_ptr(), None, None, 0,
                           cnt.data_ptr(), _stream())
    assert rc != 0
    with pytest.raises(RuntimeError):
        _lib.check(rc, "eda_wcolsum_f32")
    # more than 4 weight columns
    rc = L.eda_wcolsum_f32(x.data_ptr(), 64, 288, 290, x.data_ptr(), 290, 5, out.data_ptr(), None, None, 0,
                           cnt.data_ptr(), _stream())
    assert rc != 0
    # wgrad: M not a multiple of 4
    rc = L.eda_wgrad_f32(x.data_ptr(), 290, x.data_ptr(), 290, 64, 6, 288, out.data_ptr(), None, None, 0, _stream())
    assert rc != 0
    # colsum: workspace too small for a split reduction
    big = torch.zeros(100000, 64, device="cuda")
    o = torch.zeros(64, device="cuda")
    rc = L.eda_colsum_f32(big.data_ptr(), 100000, 64, 64, o.data_ptr(), None, 0, cnt.data_ptr(), _stream())
    assert rc != 0


def test_fused_ln_zero_rows():
    from eda_amd.fused_ln import add_dropout_layer_norm
    norm = torch.nn.LayerNorm(288).cuda()
    bias = torch.zeros(288, device="cuda", requires_grad=True)
    x = torch.empty(0, 288, device="cuda", requires_grad=True)
    y = torch.empty(0, 288, device="cuda", requires_grad=True)
    out = add_dropout_layer_norm(x, y, norm, 0.1, True, 3, y_bias=bias)
    assert out.shape == (0, 288)
    out.sum().backward()
    assert norm.weight.grad.abs().sum().item() == 0 and bias.grad.abs().sum().item() == 0


def test_device_copy_kernel():
    """The achievable-HBM yardstick of bench.py copies exactly (incl. a size that is not a multiple
    of the grid stride) and rejects unaligned sizes."""
    from eda_amd import ext
    for n in (4, 1024 * 1024 + 4, 3 * 1024 * 1024):
        src = torch.randn(n, device="cuda")
        dst = torch.full((n,), -1.0, device="cuda")
        ext.device_copy(src, dst)
        assert torch.equal(src, dst)
    with pytest.raises(RuntimeError, match="multiple of 4"):
        ext.device_copy(torch.zeros(6, device="cuda"), torch.zeros(6, device="cuda"))


def test_fps_is_exact_and_reports_status_while_another_stream_keeps_the_cus_busy(oracle):
    """The cluster FPS kernel needs the 13 workgroups of a scene co-resident (they hand candidates to
    each other).  With a second stream saturating the CUs with GEMMs -- the situation of an RCCL
    kernel or a side-stream branch next to the sampler -- the indices must stay bit-identical to the
    oracle and the sticky give-up flag (ext.fps_status) must stay clear; a give-up would be REPORTED
    there (and leave valid zero indices), never silent garbage."""
    import torch
    from eda_amd import ext, synthetic
    pc = synthetic.batch([11, 12, 13, 14], 50000)
    xyz_c = torch.from_numpy(pc[:, :, :3].copy())
    xyz = xyz_c.cuda()
    want = oracle.furthest_point_sampling(xyz_c, 512)
    ext.fps_status(reset=True)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    b = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(60):                      # ~100 ms of back-to-back full-chip GEMMs
            a = torch.mm(a, b).mul_(1e-2)
    outs = [ext.furthest_point_sampling(xyz, 512) for _ in range(6)]
    torch.cuda.synchronize()
    assert ext.fps_status() == 0
    for o in outs:
        assert (o.cpu() == want).all()
    assert bool(((outs[0] >= 0) & (outs[0] < 50000)).all())


def test_fps_give_up_is_repaired_on_the_device(oracle, monkeypatch):
    """Default policy (EDA_FPS_AUTO): behind the cluster kernels runs the bucket sampler, gated on the call's give-up
    flag.  A cluster launch that was not co-resident (simulated: EDA_FPS_TEST_GIVEUP=1 leaves what such a launch leaves
    -- flag set, zero indices -- without the ~1 s spin limit) is repaired before the call's stream work ends: exact
    indices, sticky status still clear, the repair counted.  Without a give-up the gated launch is a no-op; with the
    cluster-only policy the give-up is reported through the sticky status word as before."""
    import torch
    from eda_amd import ext, synthetic
    pc = synthetic.batch([31, 32], 30000)
    xyz_c = torch.from_numpy(pc[:, :, :3].copy())
    xyz = xyz_c.cuda()
    want = oracle.furthest_point_sampling(xyz_c, 256)
    ext.fps_status(reset=True)
    r0 = ext.fps_repaired()
    assert (ext.furthest_point_sampling(xyz, 256).cpu() == want).all() and ext.fps_repaired() == r0      # no give-up: no repair
    monkeypatch.setenv("EDA_FPS_TEST_GIVEUP", "1")
    got = ext.furthest_point_sampling(xyz, 256)
    assert (got.cpu() == want).all()
    assert ext.fps_status() == 0 and ext.fps_repaired() == r0 + 1
    ext.fps_set_policy("cluster")
    try:
        got = ext.furthest_point_sampling(xyz, 256)
        assert int(got.abs().sum()) == 0 and ext.fps_status() == 1          # reported, valid (zero) indices
    finally:
        ext.fps_set_policy("auto")
        ext.fps_status(reset=True)
    monkeypatch.delenv("EDA_FPS_TEST_GIVEUP")
    ext.fps_set_policy("bucket")
    try:
        assert (ext.furthest_point_sampling(xyz, 256).cpu() == want).all()
    finally:
        ext.fps_set_policy("auto")
