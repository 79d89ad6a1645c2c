"""Input-gradient products with a second term in their epilogue (eda_linear_addend_ws_f32 / eda_linear_dgrad_addend_ws_f32)
and their use: the residual gradient of a post-norm block rides in the attention / feed-forward branch's input-gradient
product instead of an accumulation launch (attention.ResidualLink; models/encoder_decoder_layers.py:87-105, 231-245 in
the reference, whose backward adds the two terms in autograd's own element-wise kernel).  The products are BITWISE equal to product + add; the
model-level gradients are the same sums up to the association where three terms meet."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# (R, contraction, columns): 32 x 32 / 96-wide-chunk launches, 32 x 96 tiles, the split contraction, ragged rows, the
# element-wise fallback (columns not a multiple of 4)
SHAPES = [(2048, 288, 288), (640, 288, 288), (8192, 288, 288), (8192, 576, 288), (640, 3456, 288), (1056, 3456, 288),
          (2049, 288, 576), (257, 64, 48), (100, 131, 7), (1, 4, 4)]


@pytest.mark.parametrize("R,K,N", SHAPES)
def test_addend_products_equal_product_then_add(R, K, N):
    from eda_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(R + 3 * K + 7 * N)
    x = torch.randn(R, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g)
    b = torch.randn(N, device="cuda", generator=g)
    t = torch.randn(R, N, device="cuda", generator=g)
    for bias in (None, b):
        two = gemm.linear_fwd(x, w, bias).add_(t)
        one = gemm.linear_addend(x, w, t, bias=bias)
        assert torch.equal(one, two)
        buf = t.clone()                                    # in place: the addend is the output
        assert gemm.linear_addend(x, w, buf, bias=bias, out=buf) is buf and torch.equal(buf, two)
    # input-gradient form: dy (R, N) against W (N, K) -> (R, K)
    dy = torch.randn(R, N, device="cuda", generator=g)
    t2 = torch.randn(R, K, device="cuda", generator=g)
    two = gemm.linear_dgrad(dy, w).add_(t2)
    assert torch.equal(gemm.linear_dgrad(dy, w, addend=t2), two)
    buf = t2.clone()
    gemm.linear_dgrad(dy, w, out=buf, addend=buf)
    assert torch.equal(buf, two)


def test_addend_with_the_transposed_shadow_and_strided_rows():
    from eda_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(5)
    dy = torch.randn(2048, 864, device="cuda", generator=g)[:, 288:576]          # a column slice: row stride 864
    w = torch.randn(288, 288, device="cuda", generator=g)
    big = torch.randn(2048, 576, device="cuda", generator=g)
    t = big[:, :288]                                                              # strided addend
    two = gemm.linear_dgrad(dy, w) + t
    assert torch.equal(gemm.linear_dgrad(dy, w, addend=t), two)
    assert torch.equal(gemm.linear_addend(dy, w.t().contiguous(), t), gemm.linear_fwd(dy, w.t().contiguous()) + t)


def _encoder_runs(butd):
    """The same train-mode BiEncoder (its Dropout salts are per module object) run with the links off, then on."""
    import model_fixtures as MF
    from eda_amd import attention
    from eda_amd import encoder_decoder_layers as EDL
    dev = torch.device("cuda", 0)
    d, B, V, L, D = 288, 2, 96, 12, 20
    pos = MF.make_feats(11, B, V, d, scale=0.5).to(dev)
    vmask = torch.zeros(B, V, dtype=torch.bool, device=dev)
    tmask = MF.make_mask(1, B, L, min_valid=3).to(dev)
    dmask = MF.make_mask(2, B, D, min_valid=2).to(dev) if butd else None
    layer = EDL.BiEncoderLayer(d, dropout=0.1, activation="relu", n_heads=8, dim_feedforward=256,
                               self_attend_lang=True, self_attend_vis=True, use_butd_enc_attn=butd)
    enc = EDL.BiEncoder(layer, 3)
    MF.fill_det_state(enc, seed=20)
    enc.train().to(dev)
    runs = []
    for link in (False, True):
        os.environ["EDA_RESIDUAL_LINK"] = "1" if link else "0"
        vis = MF.make_feats(10, B, V, d).to(dev).requires_grad_(True)
        text = MF.make_feats(12, B, L, d).to(dev).requires_grad_(True)
        det = MF.make_feats(13, B, D, d).to(dev).requires_grad_(True) if butd else None
        enc.zero_grad(set_to_none=True)
        attention.set_dropout_counter(dev, 1234)               # the same Dropout masks in both runs
        before = attention.ResidualLink.taken
        vo, to = enc(vis, pos, vmask, text, tmask, {}, detected_feats=det, detected_mask=dmask)
        loss = (vo * MF.make_feats(14, *vo.shape).to(dev)).sum() + (to * MF.make_feats(15, *to.shape).to(dev)).sum()
        loss.backward()
        out = [vo.detach(), to.detach(), vis.grad, text.grad] + ([det.grad] if butd else [])
        out += [p.grad.clone() for _, p in sorted(enc.named_parameters()) if p.grad is not None]
        runs.append((out, attention.ResidualLink.taken - before))
    return runs


@pytest.mark.parametrize("butd", [False, True])
def test_encoder_gradients_with_residual_links_equal_autograd_accumulation(butd):
    """Train-mode BiEncoder (3 layers, Dropout on): outputs bitwise, every input and parameter gradient equal to fp32 rounding
    with the links on and off; nine blocks hand their residual gradient over (self_vis, self_lang, text <- points per layer)."""
    try:
        (ref, n0), (got, n1) = _encoder_runs(butd)
    finally:
        os.environ.pop("EDA_RESIDUAL_LINK", None)
    assert n0 == 0 and n1 == (12 if butd else 9), (n0, n1)        # (+ points <- boxes per layer with the detector branch)
    assert len(ref) == len(got) and len(ref) > 50
    # forward untouched; a gradient where two terms meet is the same sum (fp32 addition commutes); where three meet (the text
    # features feed text <- points as query AND residual, and points <- text as key | value) the association differs:
    # autograd (kv + residual) + query, here kv + (query + residual) -- one rounding of difference, carried downstream
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
    for a, b in zip(ref[2:], got[2:]):
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-12


def test_link_is_not_armed_when_the_residual_input_needs_no_gradient():
    """x without requires_grad (a frozen trunk): the LayerNorm node keeps its ordinary return, nothing is handed over."""
    from eda_amd import attention
    from eda_amd import encoder_decoder_layers as EDL
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    layer = EDL.TransformerEncoderLayerNoFFN(288, 8, 0.1).train().to(dev)
    x = torch.randn(2, 40, 288, device=dev)
    before = attention.ResidualLink.taken
    out = layer(x, batch_first=True)
    out.sum().backward()
    assert attention.ResidualLink.taken == before
    assert layer.self_attn.in_proj_weight.grad is not None
    x2 = x.clone().requires_grad_(True)
    layer(x2, batch_first=True).sum().backward()
    assert attention.ResidualLink.taken == before + 1 and x2.grad is not None
