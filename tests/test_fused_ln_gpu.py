"""Fused LayerNorm(x + dropout(y)) (csrc/ln.hip) against a plain fp32 torch reference."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(8, 1024, 288), (8, 80, 288), (3, 7, 288), (2, 256, 64), (5, 300), (4, 16, 1000)])
def test_forward_backward_no_dropout(shape):
    from eda_amd.fused_ln import add_dropout_layer_norm
    torch.manual_seed(sum(shape))
    C = shape[-1]
    norm = torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5); norm.bias.normal_(0, 0.2)
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    y = torch.randn(*shape, device="cuda", requires_grad=True)
    w = torch.randn(*shape, device="cuda")
    out = add_dropout_layer_norm(x, y, norm, 0.1, False, 5)
    (out * w).sum().backward()
    got = [out.detach(), x.grad.clone(), y.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone()]
    for t in (x, y, norm.weight, norm.bias):
        t.grad = None
    ref = norm(x + y)
    (ref * w).sum().backward()
    exp = [ref.detach(), x.grad, y.grad, norm.weight.grad, norm.bias.grad]
    for name, g, e in zip(["out", "dx", "dy", "dgamma", "dbeta"], got, exp):
        scale = e.abs().max().item() + 1e-9
        assert (g - e).abs().max().item() <= 1e-4 * scale + 2e-6, (name, (g - e).abs().max().item(), scale)


def test_dropout_path():
    from eda_amd import attention
    from eda_amd.fused_ln import add_dropout_layer_norm, _AddDropoutLN
    torch.manual_seed(0)
    B, L, C = 4, 512, 288
    norm = torch.nn.LayerNorm(C).cuda()
    x = torch.randn(B, L, C, device="cuda")
    y = torch.randn(B, L, C, device="cuda")
    attention.dropout_state("cuda").fill_(3)
    o1 = add_dropout_layer_norm(x, y, norm, 0.1, True, 7)
    o2 = add_dropout_layer_norm(x, y, norm, 0.1, True, 7)
    assert torch.equal(o1, o2)                                   # same step + site -> same mask
    assert not torch.equal(o1, add_dropout_layer_norm(x, y, norm, 0.1, True, 8))
    attention.advance_dropout_state("cuda")
    assert not torch.equal(o1, add_dropout_layer_norm(x, y, norm, 0.1, True, 7))
    # recover the mask through the backward: with x = 0, gamma = 1 the gradient w.r.t. y is zero
    # exactly where y was dropped
    xg = torch.zeros(1, 4096, C, device="cuda")
    yg = torch.randn(1, 4096, C, device="cuda", requires_grad=True)
    out = _AddDropoutLN.apply(xg, yg, None, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), 1e-5, 0.25, 11)
    out.backward(torch.randn_like(out))
    dropped = (yg.grad == 0).float().mean().item()
    assert abs(dropped - 0.25) < 0.01, dropped
    # forward consistent with that mask: out == LN(y * mask / (1-p))
    mask = (yg.grad != 0).float()
    ref = F.layer_norm(yg.detach() * mask / 0.75, (C,))
    torch.testing.assert_close(out.detach(), ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("shape,p", [((8, 256, 288), 0.0), ((3, 77, 288), 0.1), ((2, 5, 96), 0.3)])
def test_deferred_bias(shape, p):
    """y_bias: same output as adding the bias to y beforehand; d(bias) = column sums of dy."""
    from eda_amd import attention
    from eda_amd.fused_ln import add_dropout_layer_norm
    torch.manual_seed(1 + sum(shape))
    C = shape[-1]
    norm = torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5); norm.bias.normal_(0, 0.2)
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    y = torch.randn(*shape, device="cuda", requires_grad=True)
    bias = torch.randn(C, device="cuda", requires_grad=True)
    w = torch.randn(*shape, device="cuda")
    attention.dropout_state("cuda").fill_(9)
    out = add_dropout_layer_norm(x, y, norm, p, True, 21, y_bias=bias)
    (out * w).sum().backward()
    got = [out.detach(), x.grad.clone(), y.grad.clone(), bias.grad.clone(), norm.weight.grad.clone(),
           norm.bias.grad.clone()]
    for t in (x, y, bias, norm.weight, norm.bias):
        t.grad = None
    # the dropout hash depends only on (step, site, element index): pre-adding the bias must give
    # the identical mask and hence the identical result
    yb = (y + bias)
    ref = add_dropout_layer_norm(x, yb, norm, p, True, 21)
    (ref * w).sum().backward()
    exp = [ref.detach(), x.grad, y.grad, bias.grad, norm.weight.grad, norm.bias.grad]
    for name, g, e in zip(["out", "dx", "dy", "dbias", "dgamma", "dbeta"], got, exp):
        scale = e.abs().max().item() + 1e-9
        assert (g - e).abs().max().item() <= 1e-4 * scale + 2e-6, (name, (g - e).abs().max().item(), scale)
    assert torch.equal(got[0], exp[0])


@pytest.mark.parametrize("R,C", [(8192, 576), (2048, 288), (640, 864), (2048, 3), (1, 288), (77, 130), (100000, 64),
                                 (5, 1000)])
def test_colsum(R, C):
    """csrc/colsum.hip against an fp64 column sum; repeated calls reuse the self-resetting counters."""
    from eda_amd.nn_utils import colsum
    torch.manual_seed(R + C)
    x = torch.randn(R, C, device="cuda")
    exp = x.double().sum(0)
    for _ in range(3):
        got = colsum(x)
        assert (got.double() - exp).abs().max().item() <= 1e-5 * (x.abs().double().sum(0).max().item() + 1)
    assert torch.equal(colsum(x), got)                       # deterministic order
    # strided rows (a column slice of a wider matrix) and an `out` slice
    wide = torch.randn(R, C + 8, device="cuda")
    out = torch.zeros(C + 4, device="cuda")
    colsum(wide[:, 4:4 + C], out=out[2:2 + C])
    e2 = wide[:, 4:4 + C].double().sum(0)
    assert (out[2:2 + C].double() - e2).abs().max().item() <= 1e-5 * (wide.abs().double().sum(0).max().item() + 1)
    assert out[:2].abs().sum().item() == 0 and out[2 + C:].abs().sum().item() == 0


def test_linear_rows_matches_torch():
    from eda_amd.nn_utils import linear_rows
    torch.manual_seed(3)
    x = torch.randn(8, 256, 288, device="cuda", requires_grad=True)
    W = torch.randn(64, 288, device="cuda", requires_grad=True)
    b = torch.randn(64, device="cuda", requires_grad=True)
    w = torch.randn(8, 256, 64, device="cuda")
    got = torch.autograd.grad((linear_rows(x, W, b) * w).sum(), [x, W, b])
    exp = torch.autograd.grad((F.linear(x, W, b) * w).sum(), [x, W, b])
    torch.testing.assert_close(linear_rows(x, W, b), F.linear(x, W, b), rtol=1e-5, atol=1e-5)
    for g, e in zip(got, exp):
        torch.testing.assert_close(g, e, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(8, 256, 64), (3, 80, 64), (5, 7, 288), (2, 3, 1000), (4, 1)])
def test_l2_normalize_matches_torch(shape):
    """nn_utils.l2_normalize == F.normalize(p=2, dim=-1) forward and backward, incl. all-zero rows
    (clamped at eps) and a row length that is not a multiple of 64."""
    import torch.nn.functional as F
    from eda_amd.nn_utils import l2_normalize
    torch.manual_seed(0)
    x = torch.randn(*shape, device="cuda")
    x.view(-1, shape[-1])[0].zero_()                      # a clamped row
    w = torch.randn(*shape, device="cuda")
    a = x.clone().requires_grad_(True)
    b = x.clone().requires_grad_(True)
    ya = l2_normalize(a)
    yb = F.normalize(b, p=2, dim=-1)
    torch.testing.assert_close(ya, yb, rtol=1e-6, atol=1e-7)
    (ya * w).sum().backward()
    (yb * w).sum().backward()
    # the clamped row: torch gives dy / eps there as well
    torch.testing.assert_close(a.grad, b.grad, rtol=1e-5, atol=1e-6 * float(b.grad.abs().max()))


@pytest.mark.parametrize("use", ["both", "only_pos", "only_out"])
def test_add_dropout_layer_norm_with_pos_output(use):
    """(out, out + pos) from one launch == the torch composition, forward and every input gradient,
    whichever of the two outputs the consumer uses."""
    from eda_amd.fused_ln import add_dropout_layer_norm
    torch.manual_seed(0)
    B, L, C = 3, 37, 288
    norm = torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        norm.weight.normal_(1.0, 0.2); norm.bias.normal_(0, 0.2)
    ref = torch.nn.LayerNorm(C).cuda()
    ref.load_state_dict(norm.state_dict())
    base = [torch.randn(B, L, C, device="cuda") for _ in range(3)]
    bias0 = torch.randn(C, device="cuda")
    w1, w2 = torch.randn(B, L, C, device="cuda"), torch.randn(B, L, C, device="cuda")

    def run(fused):
        x, y, pos = (t.clone().requires_grad_(True) for t in base)
        bias = bias0.clone().requires_grad_(True)
        if fused:
            out, outp = add_dropout_layer_norm(x, y, norm, 0.1, False, 7, y_bias=bias, pos=pos)
            mod = norm
        else:
            out = ref(x + (y + bias))
            outp = out + pos
            mod = ref
        loss = 0.0
        if use in ("both", "only_out"):
            loss = loss + (out * w1).sum()
        if use in ("both", "only_pos"):
            loss = loss + (outp * w2).sum()
        loss.backward()
        return out, outp, [x.grad, y.grad, pos.grad, bias.grad, mod.weight.grad, mod.bias.grad]

    oa, pa, ga = run(True)
    ob, pb, gb = run(False)
    torch.testing.assert_close(oa, ob, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(pa, pb, rtol=1e-5, atol=1e-5)
    for a, b in zip(ga, gb):
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0
        else:
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))


def _close(a, b, tol=2e-5):
    scale = float(b.abs().max()) + 1e-12
    return float((a - b).abs().max()) <= tol * scale


@pytest.mark.parametrize("R,K", [(2048, 288), (640, 288), (1000, 256), (8192, 288), (37, 32), (4100, 576)])
@pytest.mark.parametrize("p,with_pos", [(0.0, False), (0.1, True)])
def test_linear_layer_in_the_layer_norm_launch(R, K, p, with_pos, monkeypatch):
    """eda_linear_add_dropout_ln_fwd_f32 (the out-projection / second FFN linear inside the residual LayerNorm's launch)
    against the two-launch composition of the same kernels' siblings: same Dropout mask (same hash), so outputs and every
    gradient agree to fp32 rounding; and against torch in eval mode."""
    from eda_amd import fused_ln
    C = 288
    g = torch.Generator(device="cuda").manual_seed(R + K)
    base = dict(inp=torch.randn(R, K, device="cuda", generator=g), W=torch.randn(C, K, device="cuda", generator=g) * K ** -0.5,
                b=torch.randn(C, device="cuda", generator=g), x=torch.randn(R, C, device="cuda", generator=g),
                pos=torch.randn(R, C, device="cuda", generator=g))
    norm0 = torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        norm0.weight.uniform_(0.5, 1.5, generator=g); norm0.bias.normal_(generator=g)
    w1, w2 = torch.randn(R, C, device="cuda", generator=g), torch.randn(R, C, device="cuda", generator=g)

    def run(fused):
        monkeypatch.setenv("EDA_FUSED_LINEAR_LN", "1" if fused else "0")
        t = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        norm = torch.nn.LayerNorm(C).cuda()
        norm.load_state_dict(norm0.state_dict())
        assert fused_ln.fuses_linear(t["x"], norm, K) == fused
        res = fused_ln.linear_add_dropout_layer_norm(t["inp"], t["W"], t["b"], t["x"], norm, p, True, 77,
                                                     pos=t["pos"] if with_pos else None)
        out, out_pos = res if with_pos else (res, None)
        loss = (out * w1).sum() + ((out_pos * w2).sum() if with_pos else 0.0)
        loss.backward()
        grads = {k: v.grad for k, v in t.items() if v.grad is not None}
        grads["gamma"], grads["beta"] = norm.weight.grad, norm.bias.grad
        return out.detach(), (out_pos.detach() if with_pos else None), grads

    o1, op1, g1 = run(True)
    o0, op0, g0 = run(False)
    assert _close(o1, o0) and (not with_pos or _close(op1, op0))
    assert set(g1) == set(g0)
    for k in g0:
        assert _close(g1[k], g0[k], 5e-5), k
    if p == 0.0:
        with torch.no_grad():
            ref = F.layer_norm(base["x"] + F.linear(base["inp"], base["W"], base["b"]), (C,), norm0.weight, norm0.bias, norm0.eps)
        assert _close(o1, ref)


@pytest.mark.parametrize("R,K", [(2048, 288), (1030, 256), (2000, 2048), (17, 288)])
def test_split_contraction_of_the_layer_norm_launch_is_reproducible_and_close_to_unsplit(R, K, monkeypatch):
    """Up to 2048 rows two workgroups share a 16-row block's contraction and the last one to arrive adds the partial tiles in
    slice order before the LayerNorm epilogue (eda_linear_add_dropout_ln_fwd_ws_f32): 20 launches are bit-identical, the result
    is the unsplit launch's up to the association of one fp32 sum, and a replayed graph returns the same bits."""
    from eda_amd import _lib, fused_ln
    C = 288
    assert _lib.lib().eda_linear_add_dropout_ln_workspace_bytes(R, K, C) > 0
    g = torch.Generator(device="cuda").manual_seed(R * 7 + K)
    inp = torch.randn(R, K, device="cuda", generator=g); W = torch.randn(C, K, device="cuda", generator=g) * K ** -0.5
    b = torch.randn(C, device="cuda", generator=g); x = torch.randn(R, C, device="cuda", generator=g)
    gamma = torch.rand(C, device="cuda", generator=g) + 0.5; beta = torch.randn(C, device="cuda", generator=g)
    pos = torch.randn(R, C, device="cuda", generator=g)

    def run():
        out, out_pos, z, stats = fused_ln._linear_ln_forward(inp, W, b, x, gamma, beta, 1e-5, 0.1, 5, pos)
        return out, out_pos, z, stats
    first = run()
    for _ in range(20):
        for a_, b_ in zip(run(), first):
            assert torch.equal(a_, b_)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            captured = run()
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
    for a_, b_ in zip(captured, first):
        assert torch.equal(a_, b_)
    monkeypatch.setenv("EDA_GEMM_SPLITK", "0")
    assert _lib.lib().eda_linear_add_dropout_ln_workspace_bytes(R, K, C) == 0
    for a_, b_ in zip(run(), first):
        assert _close(a_, b_, 1e-5)


@pytest.mark.parametrize("R", [2048, 640, 1030])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_ffn_block_as_one_node(R, p, monkeypatch):
    """_ffn_residual_norm through _FFNAddDropoutLN (two launches) against the mlp_chain + residual LayerNorm path."""
    from eda_amd import encoder_decoder_layers as EDL
    C, Fd = 288, 256
    g = torch.Generator(device="cuda").manual_seed(R)
    ffn0 = EDL._ffn(C, Fd, p).cuda()
    norm0 = torch.nn.LayerNorm(C).cuda()
    x0 = torch.randn(8, R // 8 if R % 8 == 0 else R, C, device="cuda", generator=g) if R % 8 == 0 else torch.randn(1, R, C, device="cuda", generator=g)
    w = torch.randn_like(x0)

    def run(fused):
        monkeypatch.setenv("EDA_FUSED_LINEAR_LN", "1" if fused else "0")
        ffn, norm = EDL._ffn(C, Fd, p).cuda(), torch.nn.LayerNorm(C).cuda()
        ffn.load_state_dict(ffn0.state_dict()); norm.load_state_dict(norm0.state_dict())
        x = x0.clone().requires_grad_(True)
        out = EDL._ffn_residual_norm(x, ffn, norm, True, 4242)
        (out * w).sum().backward()
        return out.detach(), [x.grad] + [q.grad for q in ffn.parameters()] + [q.grad for q in norm.parameters()]

    o1, g1 = run(True)
    o0, g0 = run(False)
    assert _close(o1, o0)
    for a, b in zip(g1, g0):
        assert _close(a, b, 5e-5)
