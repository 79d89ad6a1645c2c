"""The repo's own fp32-MFMA row GEMMs (csrc/gemm.hip) against an fp64 torch reference.

Tolerance: fp32 accumulation over K terms; the bound used is 1e-5 * sqrt(K) * |x|.|w| row/col
norms (an fp32 dot product's forward error is ~ K*eps*sum|x_k w_k| worst case, sqrt(K)*eps typical),
far inside the north star's 1e-4 relative for activations.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # (R, K, N)
    (1, 4, 4), (7, 3, 5), (64, 6, 64), (100, 131, 128), (333, 259, 128), (2048, 288, 288),
    (2048, 288, 1), (2048, 288, 3), (640, 768, 288), (2048, 288, 864), (1000, 64, 64),
    (4096, 512, 256), (130, 288, 576), (8192, 256, 288), (20000, 128, 256), (70000, 64, 128),
    # the 256/288-wide family at ragged row counts
    (2048, 288, 256), (2049, 288, 576), (8200, 288, 288), (8192, 288, 864), (300, 256, 288), (1056, 576, 288),
    (2048, 864, 288), (5000, 288, 64), (40000, 288, 288), (257, 288, 48),
]


def _ref(x, w, b, relu):
    y = x.double() @ w.double().t()
    if b is not None:
        y = y + b.double()
    return y.relu() if relu else y


def _tol(x, w, K):
    return 2e-6 * (K ** 0.5) * (x.abs().double() @ w.abs().double().t()).clamp_min(1e-30) + 1e-30


@pytest.mark.parametrize("R,K,N", SHAPES)
def test_linear_fwd(R, K, N):
    from eda_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(R * 131 + K * 7 + N)
    x = torch.randn(R, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g)
    b = torch.randn(N, device="cuda", generator=g)
    for bias, relu in ((None, False), (b, False), (b, True)):
        y = gemm.linear_fwd(x, w, bias, relu)
        ref = _ref(x, w, bias, relu)
        err = (y.double() - ref).abs()
        assert bool((err <= _tol(x, w, K) + 1e-6 * ref.abs()).all()), float(err.max())


@pytest.mark.parametrize("R,K,N", SHAPES)
def test_linear_dgrad(R, K, N):
    from eda_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(R * 17 + K * 3 + N)
    dy = torch.randn(R, N, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g)
    dx = gemm.linear_dgrad(dy, w)
    ref = dy.double() @ w.double()
    err = (dx.double() - ref).abs()
    tol = 2e-6 * (N ** 0.5) * (dy.abs().double() @ w.abs().double()) + 1e-30
    assert bool((err <= tol + 1e-6 * ref.abs()).all()), float(err.max())


DMA_SHAPES = [(8192, 288, 576), (640, 768, 2304), (640, 3072, 768), (2048, 288, 288), (2048, 288, 256), (1000, 64, 100),
              (37, 32, 4), (1040, 288, 864), (8200, 576, 288), (33, 96, 196)]


@pytest.mark.parametrize("R,K,N", DMA_SHAPES)
def test_dma_staged_kernel_equals_register_staged_kernel(R, K, N, monkeypatch):
    """gemm_dma_kernel (global_load_lds staging, every configuration, both tile -> XCD maps come up through the shapes)
    forms the same products in the same order as gemm_rows_kernel: forward (bias, ReLU / dropout / gate epilogues) and
    dX agree bit for bit.  (Unsplit contraction: the split form adds slice partials, another association.)"""
    from eda_amd import _lib, gemm
    monkeypatch.setenv("EDA_GEMM_SPLITK", "0")
    monkeypatch.setenv("EDA_GEMM_B3ROWS", "0")          # (8192 rows: the fp32 kernels under comparison, not the bf16 x 3 one)
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(R + K + N)
    x = torch.randn(R, K, device="cuda", generator=g); w = torch.randn(N, K, device="cuda", generator=g)
    b = torch.randn(N, device="cuda", generator=g); dy = torch.randn(R, N, device="cuda", generator=g)
    seed = torch.tensor([1234], dtype=torch.int64, device="cuda")
    gate = torch.randn(R, N, device="cuda", generator=g)

    def run():
        return (gemm.linear_fwd(x, w, b), gemm.linear_fwd(x, w, None, True), gemm.linear_dgrad(dy, w),
                gemm.linear_ex(x, w, b, relu=True, drop=(0.1, seed, 7)), gemm.linear_ex(x, w, None, gate=(gate, 1.25)))
    try:
        assert L.eda_gemm_set_dma(0) == 0
        base = run()
        for mode in (1, 2, 3, 4):
            assert L.eda_gemm_set_dma(mode) == 0
            for got, want in zip(run(), base):
                assert torch.equal(got, want), (mode, float((got - want).abs().max()))
    finally:
        L.eda_gemm_set_dma(-1)
    ref = x.double() @ w.double().t() + b.double()
    assert float((base[0].double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("R,K,N", [(2048, 288, 288), (100, 288, 96), (640, 576, 288), (8192, 288, 576), (33, 96, 36), (2048, 864, 288)])
def test_96_wide_chunks_equal_32_wide_chunks(R, K, N, monkeypatch):
    """gemm_dma_kernel with KCT = 96 (EDA_GEMM_KC96 = tile configuration): three 32-wide chunks per barrier phase, the
    same MFMA sequence -- forward (bias, ReLU / dropout / gate epilogues) and dX bit for bit equal to the kernels of
    EDA_GEMM_KC96=0, for every tile configuration the dispatch can pick."""
    from eda_amd import _lib, gemm
    monkeypatch.setenv("EDA_GEMM_SPLITK", "0")
    monkeypatch.setenv("EDA_GEMM_B3ROWS", "0")          # (8192 rows: the fp32 kernels under comparison, not the bf16 x 3 one)
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(R + K + N)
    x = torch.randn(R, K, device="cuda", generator=g); w = torch.randn(N, K, device="cuda", generator=g)
    b = torch.randn(N, device="cuda", generator=g); dy = torch.randn(R, N, device="cuda", generator=g)
    seed = torch.tensor([99], dtype=torch.int64, device="cuda")
    gate = torch.randn(R, N, device="cuda", generator=g)

    def run():
        return (gemm.linear_fwd(x, w, b), gemm.linear_fwd(x, w, None, True), gemm.linear_dgrad(dy, w),
                gemm.linear_ex(x, w, b, relu=True, drop=(0.1, seed, 7)), gemm.linear_ex(x, w, None, gate=(gate, 1.25)))
    monkeypatch.setenv("EDA_GEMM_KC96", "0"); L.eda_reload_env()
    base = run()
    for cfg in range(1, 9):
        monkeypatch.setenv("EDA_GEMM_KC96", str(cfg)); L.eda_reload_env()
        for i, (got, want) in enumerate(zip(run(), base)):
            assert torch.equal(got, want), (cfg, i, float((got - want).abs().max()))
    ref = x.double() @ w.double().t() + b.double()
    assert float((base[0].double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_strided_operands_and_outputs():
    """Column views of packed projection buffers (row stride 864) as inputs and outputs."""
    from eda_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(5)
    big = torch.randn(512, 864, device="cuda", generator=g)
    x = big[:, 288:576]
    w = torch.randn(3 * 288, 288, device="cuda", generator=g)
    out = torch.full((512, 864), float("nan"), device="cuda")
    gemm.linear_fwd(x, w[288:576], None, False, out=out[:, 576:])
    ref = x.double() @ w[288:576].double().t()
    assert torch.allclose(out[:, 576:].double(), ref, rtol=1e-5, atol=1e-3)
    assert torch.isnan(out[:, :576]).all()
    dx = gemm.linear_dgrad(big[:, 576:], w[576:])
    assert torch.allclose(dx.double(), big[:, 576:].double() @ w[576:].double(), rtol=1e-5, atol=1e-3)


def test_transposed_shadow_input_gradients_equal_the_plain_form():
    """eda_amd/wt_shadow.py: W^T of a list of weights by ONE launch; linear_dgrad through the shadow (the forward's
    GEMM form on W^T) equals the plain form, for whole weights and for row ranges of a packed in-projection."""
    from eda_amd import gemm, wt_shadow
    g = torch.Generator(device="cuda").manual_seed(5)
    flat = torch.randn(864 * 288 + 288 * 256 + 64 * 288 + 40 * 20 + 16, device="cuda", generator=g)
    w_in = flat[:864 * 288].view(864, 288)
    w_ffn = flat[864 * 288:864 * 288 + 288 * 256].view(288, 256)
    o = 864 * 288 + 288 * 256
    w_c = flat[o:o + 64 * 288].view(64, 288, 1)                     # a 1x1 convolution's weight
    w_odd = flat[o + 64 * 288:o + 64 * 288 + 800].view(40, 20)
    sh = wt_shadow.TransposedShadow([w_in, w_ffn, w_c, w_odd, None])
    assert len(sh) == 4
    sh.refresh()
    for w in (w_in, w_ffn, w_c.squeeze(-1), w_odd):
        assert torch.equal(sh.lookup(w), w.t())
    assert sh.lookup(torch.randn(288, 288, device="cuda")) is None
    cases = [(w_in, 0, 864), (w_in, 0, 576), (w_in, 576, 864), (w_in, 288, 576), (w_ffn, 0, 288), (w_c.squeeze(-1), 0, 64)]
    for w, lo, hi in cases:
        ws = w[lo:hi]
        dy = torch.randn(2048, hi - lo, device="cuda", generator=g)
        plain = gemm.linear_dgrad(dy, ws)
        wt_shadow.active = sh
        try:
            via = gemm.linear_dgrad(dy, ws)
        finally:
            wt_shadow.active = None
        ref = dy.double() @ ws.double()
        tol = 2e-6 * (ws.shape[0] ** 0.5) * (dy.abs().double() @ ws.abs().double()) + 1e-30
        assert bool(((via.double() - ref).abs() <= tol).all())
        assert bool(((plain.double() - ref).abs() <= tol).all())
    w_in.mul_(2.0)                                                  # a stale shadow is visible until refresh()
    assert not torch.equal(sh.lookup(w_in), w_in.t())
    sh.refresh()
    assert torch.equal(sh.lookup(w_in), w_in.t())


@pytest.mark.parametrize("shadow", [False, True])
@pytest.mark.parametrize("R,dims,p", [(2048, (288, 256, 288), 0.0), (640, (288, 256, 288), 0.1), (777, (288, 288, 288, 64), 0.0),
                                      (2048, (288, 256, 288), 0.3)])
def test_mlp_chain_node_matches_the_torch_composition(R, dims, p, shadow):
    """nn_utils._MLPChain (Linear -> ReLU -> Dropout -> ... -> Linear as one node, activations in the GEMM epilogues)
    against torch: outputs and all gradients.  With p > 0 the node's own keep mask is read back from its hidden
    activation (h != 0 <=> kept and positive) and imposed on the torch composition; the kept fraction is checked."""
    from eda_amd import nn_utils, wt_shadow
    g = torch.Generator(device="cuda").manual_seed(R + len(dims))
    n = len(dims) - 1
    Ws = [(torch.randn(dims[l + 1], dims[l], device="cuda", generator=g) * dims[l] ** -0.5).requires_grad_(True) for l in range(n)]
    bs = [(torch.randn(dims[l + 1], device="cuda", generator=g) * 0.1).requires_grad_(True) for l in range(n)]
    x = torch.randn(R, dims[0], device="cuda", generator=g, requires_grad=True)
    seed = torch.full((1,), 1234567, dtype=torch.int64, device="cuda")
    layers = [(Ws[l], bs[l], l < n - 1, p if l < n - 1 else 0.0, 17 + l) for l in range(n)]
    dy = torch.randn(R, dims[-1], device="cuda", generator=g)
    sh = wt_shadow.TransposedShadow(Ws) if shadow else None
    if sh is not None:
        sh.refresh()
    y = nn_utils.mlp_chain(x, layers, True, seed=seed)
    wt_shadow.active = sh
    try:
        y.backward(dy)
    finally:
        wt_shadow.active = None
    got = [y.detach(), x.grad] + [w.grad for w in Ws] + [b.grad for b in bs]
    # torch composition in fp64 with the node's masks
    xd = x.detach().double().requires_grad_(True)
    Wd = [w.detach().double().requires_grad_(True) for w in Ws]
    bd = [b.detach().double().requires_grad_(True) for b in bs]
    h, h32 = xd, x.detach()
    for l in range(n):
        z = h @ Wd[l].t() + bd[l]
        if l < n - 1:
            # the node's own decisions (ReLU sign in ITS fp32 arithmetic, Dropout keep mask): its hidden activation
            # after layer l is non-zero exactly where the element is positive and kept
            hl = nn_utils.mlp_chain(x.detach(), layers[:l + 1], True, seed=seed)
            z32 = h32 @ Ws[l].detach().t() + bs[l].detach()
            live = hl != 0
            if p > 0:
                pos = (z32 > 1e-4)
                dropped = (pos & ~live).float().sum().item() / max(pos.float().sum().item(), 1.0)
                assert abs(dropped - p) < 0.02, dropped
            else:
                assert ((z32 > 1e-4) & ~live).sum().item() == 0 and ((z32 < -1e-4) & live).sum().item() == 0
            z = z * live.double() / (1.0 - p)
            h32 = hl
        h = z
    h.backward(dy.double())
    exp = [h.detach(), xd.grad] + [w.grad for w in Wd] + [b.grad for b in bd]
    for a, e in zip(got, exp):
        scale = e.abs().max().item() + 1e-12
        assert (a.double() - e).abs().max().item() <= 2e-4 * scale, ((a.double() - e).abs().max().item(), scale)


def test_transpose_batch_c_entry_handles_ragged_matrices():
    """eda_transpose_batch_f32 straight through the C ABI: matrices whose sides are not multiples of the 32x32 tile."""
    from eda_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(11)
    shapes = [(37, 53), (1, 1), (288, 864), (33, 32), (5, 260)]
    srcs = [torch.randn(r, c, device="cuda", generator=g) for r, c in shapes]
    dsts = [torch.full((c, r), float("nan"), device="cuda") for r, c in shapes]
    desc, tiles = [], 0
    for s_, d_, (r, c) in zip(srcs, dsts, shapes):
        desc.append([s_.data_ptr(), d_.data_ptr(), r, c, tiles])
        tiles += ((r + 31) // 32) * ((c + 31) // 32)
    dt = torch.tensor(desc, dtype=torch.int64, device="cuda")
    rc = _lib.lib().eda_transpose_batch_f32(dt.data_ptr(), len(shapes), tiles, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_transpose_batch_f32")
    for s_, d_ in zip(srcs, dsts):
        assert torch.equal(d_, s_.t())


SPLITK_SHAPES = [(640, 3072, 768), (640, 3456, 288), (1056, 3456, 288), (640, 768, 768), (130, 2304, 100), (33, 1024, 96)]


@pytest.mark.parametrize("R,K,N", SPLITK_SHAPES)
def test_split_contraction_is_reproducible_and_matches_fp64(R, K, N, monkeypatch):
    """Split contraction of the DMA-staged products (csrc/gemm.hip SK): few 32 x 96 tiles against a long contraction.
    Forward form with bias + GELU / gate epilogues and the input-gradient form; two calls give the same bits (the slices
    are added in slice order whoever arrives last, the ticket words are re-armed); forced slice counts agree to fp32."""
    from eda_amd import _lib, gemm
    g = torch.Generator(device="cuda").manual_seed(R + K + N)
    x = torch.randn(R, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g)
    b = torch.randn(N, device="cuda", generator=g)
    dy = torch.randn(R, N, device="cuda", generator=g)
    outs = {}
    for sk in (None, "0", "2", "5"):
        if sk is None:
            monkeypatch.delenv("EDA_GEMM_SPLITK", raising=False)
        else:
            monkeypatch.setenv("EDA_GEMM_SPLITK", sk)
        nbytes = _lib.lib().eda_linear_splitk_workspace_bytes(R, K, N)
        assert (nbytes > 0) == (sk != "0") or sk is None
        y1 = gemm.linear_fwd(x, w, b, relu=2)
        y2 = gemm.linear_fwd(x, w, b, relu=2)
        assert torch.equal(y1, y2)
        gate = torch.randn(R, N, device="cuda", generator=g)
        y3 = gemm.linear_ex(x, w, None, gate=(gate, 0.5))
        dx = gemm.linear_dgrad(dy, w)                      # contraction over N: split only where N is long
        outs[sk] = (y1, y3, dx)
        z = x.double() @ w.double().t()
        ref = torch.nn.functional.gelu(z + b.double())
        err = (y1.double() - ref).abs()
        assert bool((err <= _tol(x, w, K) + 2e-6 * ref.abs() + 1e-6).all()), (sk, float(err.max()))
        ref3 = torch.where(gate > 0, z * 0.5, torch.zeros_like(z))
        assert bool(((y3.double() - ref3).abs() <= _tol(x, w, K) + 1e-6 * ref3.abs()).all()), sk
        refd = dy.double() @ w.double()
        told = 2e-6 * (N ** 0.5) * (dy.abs().double() @ w.abs().double()) + 1e-30
        assert bool(((dx.double() - refd).abs() <= told + 1e-6 * refd.abs()).all()), sk
    torch.cuda.synchronize()
    for ws in gemm._sk_ws_cache.values():
        assert int(ws[:1024].abs().sum().item()) == 0


def test_split_contraction_input_gradient_form():
    """dX = dY W with a long contraction over the layer's outputs (the hoisted K | V projections: 3456 outputs)."""
    from eda_amd import _lib, gemm
    R, N, K = 640, 3456, 288
    assert _lib.lib().eda_linear_splitk_workspace_bytes(R, N, K) > 0
    g = torch.Generator(device="cuda").manual_seed(5)
    dy = torch.randn(R, N, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g)
    dx = gemm.linear_dgrad(dy, w)
    assert torch.equal(dx, gemm.linear_dgrad(dy, w))
    ref = dy.double() @ w.double()
    tol = 2e-6 * (N ** 0.5) * (dy.abs().double() @ w.abs().double()) + 1e-30
    assert bool(((dx.double() - ref).abs() <= tol + 1e-6 * ref.abs()).all())


B3_SHAPES = [(8192, 288, 288), (8192, 288, 576), (8192, 576, 288), (8200, 288, 864), (4097, 288, 48), (40000, 288, 288),
             (5000, 576, 32)]


@pytest.mark.parametrize("R,K,N", B3_SHAPES)
def test_many_row_products_on_the_bf16_pipe_keep_fp32_accuracy(R, K, N, monkeypatch):
    """gemm_b3_rows_kernel (>= 4096 plain rows against a 288- / 576-deep contraction: operands split into three bf16 terms,
    six v_mfma_f32_16x16x32_bf16 per step): every epilogue (bias, ReLU, Dropout, gate, addend), ragged row counts; the error
    against fp64 stays inside the fp32 kernels' bound (this file's _tol) and is not larger than 1.5 x the fp32-MFMA kernel's
    own; the Dropout mask and the ReLU / gate decisions are the fp32 kernel's wherever the pre-activation is not within
    rounding of zero."""
    from eda_amd import _lib, gemm
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(R + 5 * K + 11 * N)
    x = torch.randn(R, K, device="cuda", generator=g); w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    b = torch.randn(N, device="cuda", generator=g)
    seed = torch.tensor([99], dtype=torch.int64, device="cuda")
    gate = torch.randn(R, N, device="cuda", generator=g)
    big = torch.randn(R, K + 64, device="cuda", generator=g)
    xs = big[:, 32:32 + K]                                   # strided rows (16-byte aligned)

    def run():
        return (gemm.linear_fwd(x, w, b), gemm.linear_fwd(x, w, None, True), gemm.linear_fwd(xs, w, b),
                gemm.linear_ex(x, w, b, relu=True, drop=(0.1, seed, 7)), gemm.linear_ex(x, w, None, gate=(gate, 1.25)),
                gemm.linear_addend(x, w, gate, bias=b))
    try:
        monkeypatch.setenv("EDA_GEMM_B3ROWS", "0"); L.eda_reload_env()
        base = run()
        monkeypatch.setenv("EDA_GEMM_B3ROWS", "2"); L.eda_reload_env()          # (2: every eligible shape, not only where it wins)
        got = run()
    finally:
        monkeypatch.delenv("EDA_GEMM_B3ROWS"); L.eda_reload_env()
    ref = x.double() @ w.double().t()
    refs = (ref + b.double(), ref.relu(), xs.double() @ w.double().t() + b.double(), None, None, ref + b.double() + gate.double())
    tol = _tol(x, w, K)
    assert not torch.equal(got[0], base[0])                  # (the other kernel did run)
    for i in (0, 1, 2, 5):
        e_new = (got[i].double() - refs[i]).abs()
        e_old = (base[i].double() - refs[i]).abs()
        assert bool((e_new <= (_tol(xs, w, K) if i == 2 else tol) + 1e-6 * refs[i].abs()).all()), (i, float(e_new.max()))
        assert float(e_new.max()) <= 1.5 * float(e_old.max()) + 1e-7, (i, float(e_new.max()), float(e_old.max()))
    # Dropout / gate epilogues: same mask (a counter-based hash of the element), same values to fp32 rounding
    for i in (3, 4):
        d = (got[i] - base[i]).abs()
        flips = (got[i] == 0) != (base[i] == 0)              # a ReLU decision within rounding of zero
        assert float(flips.float().mean()) < 1e-4
        assert bool((d[~flips] <= 2 * tol[~flips].float() + 1e-6).all()), (i, float(d[~flips].max()))
