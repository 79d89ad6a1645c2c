"""Attention parity.
CPU: the torch restatement (oracle/attention_ref.py) is pinned against
torch.nn.MultiheadAttention -- the module the reference instantiates -- through
this repo's MultiheadAttention (same parameters, batch-first).
GPU: the fused HIP kernels (csrc/mha2.hip; the long-key forward csrc/mha3.hip) against that restatement: outputs and
gradients <= 1e-4 relative (fp32 MFMA), masks, ragged lengths, strided (packed
QKV) inputs, and the dropout path (keep rate, fwd/bwd mask consistency)."""
import numpy as np
import pytest
import torch

from oracle import attention_ref


def _mask(B, L, seed, min_valid=1):
    rng = np.random.default_rng(seed)
    lens = rng.integers(min_valid, L + 1, B)
    lens[0] = L
    return torch.from_numpy(np.arange(L)[None, :] >= lens[:, None])


@pytest.mark.parametrize("case", ["self", "posself", "cross"])
def test_module_matches_torch_multiheadattention(case, monkeypatch):
    from eda_amd import attention
    monkeypatch.setattr(attention, "_core", attention_ref.attention_core)
    torch.manual_seed(0)
    ref = torch.nn.MultiheadAttention(288, 8, dropout=0.1).eval()
    mine = attention.MultiheadAttention(288, 8, dropout=0.1).eval()
    with torch.no_grad():
        ref.in_proj_bias.normal_(0, 0.1); ref.out_proj.bias.normal_(0, 0.1)
    mine.load_state_dict(ref.state_dict())
    B, Lq, Lk = 3, 20, 33
    x = torch.randn(B, Lq, 288); pos = torch.randn(B, Lq, 288); mem = torch.randn(B, Lk, 288)
    if case == "self":
        q = k = v = x; mask = _mask(B, Lq, 1)
    elif case == "posself":
        q = k = x + pos; v = x; mask = None
    else:
        q = x; k = v = mem; mask = _mask(B, Lk, 2)
    exp = ref(q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1), key_padding_mask=mask)[0].transpose(0, 1)
    got = mine(q, k, v, key_padding_mask=mask, batch_first=True)[0]
    torch.testing.assert_close(got, exp, rtol=1e-5, atol=1e-5)
    got_sf = mine(q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1), key_padding_mask=mask)[0]
    torch.testing.assert_close(got_sf.transpose(0, 1), exp, rtol=1e-5, atol=1e-5)


def test_core_refuses_cpu():
    from eda_amd import attention
    with pytest.raises(RuntimeError, match="CPU not supported"):
        attention.attention_core(torch.zeros(1, 4, 288), torch.zeros(1, 4, 288), torch.zeros(1, 4, 288))


# ------------------------------------------------------------------ GPU ------
SHAPES = [  # (B, Lq, Lk, masked)
    (2, 16, 16, False), (2, 80, 80, True), (1, 1024, 1024, False), (2, 80, 1024, False),
    (2, 1024, 80, True), (2, 1024, 132, True), (2, 256, 256, False), (2, 256, 80, True),
    (3, 37, 101, True), (1, 5, 1, False), (2, 64, 65, True),
    # every count of live 16-row sub-tiles in the last tile (the tile bodies are templated on it): 2 and 3
    (2, 90, 24, False), (2, 24, 90, True), (2, 110, 175, True),
    # SR3D-shaped long utterances (BASELINE.json configs[4]: 130 tokens)
    (2, 130, 130, True), (2, 130, 1024, False), (2, 1024, 130, True), (2, 256, 130, True),
    # key-split forward (csrc/mha2.hip SPLIT: <= 256 queries against >= 512 keys): the bench shapes with masks, a ragged
    # key count (3 chunks, the last one partial), the full batch (2 splits per block instead of 4)
    (2, 256, 1024, True), (1, 200, 700, True), (8, 80, 1024, True), (8, 256, 1024, False),
]


@pytest.mark.gpu
@pytest.mark.parametrize("B,Lq,Lk,masked", SHAPES)
def test_fused_forward_backward_vs_restatement(B, Lq, Lk, masked):
    from eda_amd import attention
    torch.manual_seed(Lq * 7 + Lk)
    dev = "cuda"
    q = torch.randn(B, Lq, 288, device=dev, requires_grad=True)
    k = torch.randn(B, Lk, 288, device=dev, requires_grad=True)
    v = torch.randn(B, Lk, 288, device=dev, requires_grad=True)
    mask = _mask(B, Lk, Lq + Lk).to(dev) if masked else None
    w = torch.randn(B, Lq, 288, device=dev)
    out = attention.attention_core(q, k, v, mask, 8, 0.0, 0)
    (out * w).sum().backward()
    got = [out.detach(), q.grad.clone(), k.grad.clone(), v.grad.clone()]
    for t in (q, k, v):
        t.grad = None
    # checker in fp64 on the same device
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    exp_out = attention_ref.attention_core(qd, kd, vd, mask, 8)
    (exp_out * w.double()).sum().backward()
    exp = [exp_out.detach(), qd.grad, kd.grad, vd.grad]
    # element-wise: 1e-4 relative (the north star's bound for fp32 activations) plus an absolute term for
    # elements that cancel to ~0: an fp32 sum of n terms carries up to n*2^-24 of sum|terms|; the bound
    # 2e-6 * max|expected| is that for the ~30-term effective sums here with the tensor's own scale as the
    # term size (measured use on MI355X: <= 40 % on every shape)
    cancel = (w.abs().max() * v.abs().max()).item()      # dq / dk = P (dP - delta) K: exactly 0 for a one-hot softmax
    natural = {"out": 0.0, "dv": 0.0, "dq": cancel * k.abs().max().item(), "dk": cancel * q.abs().max().item()}
    for name, g, e in zip(["out", "dq", "dk", "dv"], got, exp):
        err = (g.double() - e).abs()
        tol = 1e-4 * e.abs() + 2e-6 * e.abs().max() + 2e-7 * natural[name] + 1e-30
        worst = (err / tol).max().item()
        assert worst <= 1.0, (name, worst, err.max().item(), e.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("B,Lq,Lk", [(2, 80, 1024), (8, 256, 1024), (1, 200, 700), (3, 64, 513)])
def test_key_split_forward_is_reproducible_and_matches_the_unsplit_kernel(B, Lq, Lk, monkeypatch):
    """The last workgroup of a block to arrive merges the splits in split order: two calls give the same bits (and the
    ticket words are back at zero for the second one); against the one-workgroup-per-block kernel (EDA_MHA2_KSPLIT=0)
    the result differs by the re-association of the online softmax only; forcing other split counts works too."""
    from eda_amd import _lib, attention
    monkeypatch.setenv("EDA_MHA4", "0")                 # (this test is about mha2.hip's key split; mha4.hip has its own below)
    torch.manual_seed(Lq + Lk)
    dev = "cuda"
    q, k, v = (torch.randn(B, L, 288, device=dev) for L in (Lq, Lk, Lk))
    mask = _mask(B, Lk, 5, min_valid=Lk // 3).to(dev)
    # (the library's own plan splits while a (scene, head, query block) grid leaves CUs idle: not at B = 8 x 256 queries)
    assert (_lib.lib().eda_mha_fwd_workspace_bytes(B, 8, Lq, Lk) > 0) == (B * 8 * ((Lq + 63) // 64) < 224)
    a = attention.attention_core(q, k, v, mask, 8, 0.1, 7)
    b = attention.attention_core(q, k, v, mask, 8, 0.1, 7)
    assert torch.equal(a, b)
    outs = {}
    for ks in ("0", "2", "3"):
        monkeypatch.setenv("EDA_MHA2_KSPLIT", ks)
        assert (_lib.lib().eda_mha_fwd_workspace_bytes(B, 8, Lq, Lk) > 0) == (ks != "0")
        outs[ks] = attention.attention_core(q, k, v, mask, 8, 0.1, 7)
    monkeypatch.delenv("EDA_MHA2_KSPLIT")
    scale = outs["0"].abs().max().item()
    for ks in ("2", "3"):
        assert (outs[ks] - outs["0"]).abs().max().item() <= 2e-6 * scale, ks
    assert (a - outs["0"]).abs().max().item() <= 2e-6 * scale
    torch.cuda.synchronize()
    for ws in attention._fwd_ws_cache.values():          # every ticket word re-armed
        assert int(ws[:64].abs().sum().item()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("B,Lq,Lk,mode", [(8, 80, 1024, "1"), (8, 130, 1024, "1"), (2, 80, 1024, "1"), (3, 64, 513, "1"),
                                          (1, 144, 700, "1"), (2, 17, 512, "1"), (5, 1, 1000, "1"), (8, 256, 1024, "2"),
                                          (1, 200, 700, "2")])
@pytest.mark.parametrize("masked", [True, False])
def test_keys_per_wave_forward_matches_the_query_per_wave_kernel(B, Lq, Lk, mode, masked, monkeypatch):
    """csrc/mha4.hip (round 6): a wave owns 32 keys and walks over all query tiles; the waves' and the key splits' (O, m, l)
    are merged in wave / split order.  Against mha2.hip's one-workgroup-per-block kernel (EDA_MHA4=0, EDA_MHA2_KSPLIT=0) the
    output and lse differ by the re-association of the softmax sums only (same dropout mask: the hash is indexed by query and
    key); two calls give the same bits whoever arrives last; the ticket words are back at zero."""
    from eda_amd import _lib, attention
    L = _lib.lib()
    torch.manual_seed(Lq * 7 + Lk)
    dev = "cuda"
    q, k, v = (torch.randn(B, n, 288, device=dev) for n in (Lq, Lk, Lk))
    m8 = _mask(B, Lk, 5, min_valid=Lk // 3).to(dev).contiguous().view(torch.uint8) if masked else None

    def fwd():
        out, lse = torch.full((B, Lq, 288), float("nan"), device=dev), torch.full((B, 8, Lq), float("nan"), device=dev)
        attention._mha_fwd_call(q, k, v, m8, B, 8, Lq, Lk, 36, 0.1, 7, out, lse)
        return out, lse
    monkeypatch.setenv("EDA_MHA4", mode)
    assert L.eda_mha_fwd_workspace_bytes(B, 8, Lq, Lk) >= B * 8 * ((Lk + 255) // 256) * ((Lq + 15) // 16) * 4096
    o1, l1 = fwd()
    o2, l2 = fwd()
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    torch.cuda.synchronize()
    for ws in attention._fwd_ws_cache.values():
        assert int(ws[:min(B * 8, ws.numel())].abs().sum().item()) == 0        # (one ticket word per (scene, head))
    monkeypatch.setenv("EDA_MHA4", "0")
    monkeypatch.setenv("EDA_MHA2_KSPLIT", "0")
    o0, l0 = fwd()
    assert torch.isfinite(o0).all() and torch.isfinite(l0).all()
    assert (o1 - o0).abs().max().item() <= 2e-6 * o0.abs().max().item()
    assert (l1 - l0).abs().max().item() <= 2e-6 * l0.abs().max().item()


@pytest.mark.gpu
def test_keys_per_wave_forward_with_all_keys_of_a_scene_masked_gives_nan_like_the_reference(monkeypatch):
    from eda_amd import attention
    dev = "cuda"
    torch.manual_seed(1)
    q, k, v = (torch.randn(2, n, 288, device=dev) for n in (80, 1024, 1024))
    mask = torch.zeros(2, 1024, dtype=torch.bool, device=dev)
    mask[1] = True
    monkeypatch.setenv("EDA_MHA4", "1")
    out = attention.attention_core(q, k, v, mask, 8, 0.0, 3)
    assert torch.isfinite(out[0]).all() and torch.isnan(out[1]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("B,Lq,Lk", [(8, 1024, 1024), (8, 80, 1024), (8, 1024, 80), (8, 256, 132), (2, 300, 700), (1, 256, 256),
                                     (3, 1024, 16)])
@pytest.mark.parametrize("mode", ["1", "0", None])
def test_split_backward_merges_in_the_launch_reproducibly(B, Lq, Lk, mode, monkeypatch):
    """Round 6: the key blocks' dQ partials and the query splits' dK | dV partials can be merged by the LAST workgroup of a
    range inside the backward launch (ticket + write-through partials; EDA_MHA2_BWD_MERGE=1) instead of by a second launch
    (=0); unset, the library decides by the size of a range (small ranges in the launch).  In every mode: two calls give the
    same bits whoever arrives last; the ticket words are back at zero; the legacy entry point (tickets at the head of its
    per-call scratch, zeroed by a fill kernel) gives the same bits as the persistent-ticket entry point; and the modes agree
    bit for bit with each other (both sum the same partials in split order)."""
    from eda_amd import _lib, attention
    L = _lib.lib()
    if mode is not None:
        monkeypatch.setenv("EDA_MHA2_BWD_MERGE", mode)
    torch.manual_seed(Lq * 3 + Lk)
    dev = "cuda"
    q, k, v = (torch.randn(B, n, 288, device=dev) for n in (Lq, Lk, Lk))
    mask = _mask(B, Lk, 5, min_valid=max(1, Lk // 3)).to(dev)
    dout = torch.randn(B, Lq, 288, device=dev)
    assert L.eda_mha_bwd_workspace_bytes(B, 8, Lq, Lk) > 0         # every shape of this list is split somewhere
    if mode is not None:
        assert (L.eda_mha_bwd_ticket_bytes(B, 8, Lq, Lk) > 0) == (mode == "1")

    def run():
        qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
        out = attention.attention_core(qq, kk, vv, mask, 8, 0.1, 7)
        out.backward(dout)
        return out.detach(), qq.grad, kk.grad, vv.grad
    o1, dq1, dk1, dv1 = run()
    o2, dq2, dk2, dv2 = run()
    assert torch.equal(dq1, dq2) and torch.equal(dk1, dk2) and torch.equal(dv1, dv2)
    torch.cuda.synchronize()
    for tk in attention._bwd_tk_cache.values():
        assert int(tk.abs().sum().item()) == 0
    # the legacy entry point: per-call scratch, tickets zeroed by a fill launch
    lse = torch.empty(B, 8, Lq, device=dev)
    out = torch.empty(B, Lq, 288, device=dev)
    m8 = mask.contiguous().view(torch.uint8)
    seed = attention.dropout_state(torch.device(dev))
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.eda_mha_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), q.stride(1), k.stride(0), k.stride(1),
                             v.stride(0), v.stride(1), m8.data_ptr(), B, 8, Lq, Lk, 36, 36 ** -0.5, 0.1, seed.data_ptr(), 7,
                             out.data_ptr(), lse.data_ptr(), 0, st), "fwd")
    dq3, dk3, dv3 = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
    nws = L.eda_mha_bwd_workspace_bytes(B, 8, Lq, Lk)
    ws = torch.full((nws // 4,), float("nan"), device=dev)        # (poisoned: the call must zero its own ticket area)
    _lib.check(L.eda_mha_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), q.stride(1), k.stride(0), k.stride(1),
                             v.stride(0), v.stride(1), m8.data_ptr(), B, 8, Lq, Lk, 36, 36 ** -0.5, 0.1, seed.data_ptr(), 7,
                             out.data_ptr(), lse.data_ptr(), dout.data_ptr(), dout.stride(0), dout.stride(1), None,
                             dq3.data_ptr(), dk3.data_ptr(), dv3.data_ptr(), dq3.stride(0), dq3.stride(1), dk3.stride(0),
                             dk3.stride(1), dv3.stride(0), dv3.stride(1), ws.data_ptr(), nws, 0, st), "bwd")
    # ... against the persistent-ticket entry point on the SAME saved forward (out, lse)
    dq4, dk4, dv4 = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
    attention._mha_bwd_call(q, k, v, m8, B, 8, Lq, Lk, 36, 0.1, seed, 7, out, lse, dout, dq4, dk4, dv4, 0)
    assert torch.equal(dq3, dq4) and torch.equal(dk3, dk4) and torch.equal(dv3, dv4)
    # (the autograd path above may have taken the key-split forward: same numbers up to the online softmax's re-association)
    for a, b in ((dq1, dq4), (dk1, dk4), (dv1, dv4)):
        assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()
    # the other merge mode on the same saved forward: the same partials summed in the same order
    if mode is not None:
        monkeypatch.setenv("EDA_MHA2_BWD_MERGE", "0" if mode == "1" else "1")
        dq5, dk5, dv5 = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
        attention._mha_bwd_call(q, k, v, m8, B, 8, Lq, Lk, 36, 0.1, seed, 7, out, lse, dout, dq5, dk5, dv5, 0)
        assert torch.equal(dq5, dq4) and torch.equal(dk5, dk4) and torch.equal(dv5, dv4)


@pytest.mark.gpu
def test_fused_strided_packed_inputs():
    """q/k/v as column slices of one packed projection output (what the module passes)."""
    from eda_amd import attention
    torch.manual_seed(3)
    B, L = 2, 200
    packed = torch.randn(B, L, 864, device="cuda", requires_grad=True)
    q, k, v = packed.split(288, dim=-1)
    assert not q.is_contiguous()
    out = attention.attention_core(q, k, v, None, 8, 0.0, 0)
    out.pow(2).sum().backward()
    g = packed.grad.clone()
    packed.grad = None
    exp = attention_ref.attention_core(*packed.double().split(288, dim=-1), None, 8)
    exp.pow(2).sum().backward()
    torch.testing.assert_close(out.double(), exp, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(g.double(), packed.grad.double(), rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_fused_dropout_statistics_and_gradient_consistency():
    from eda_amd import attention
    dev = "cuda"
    torch.manual_seed(5)
    B, Lq, Lk = 2, 128, 192
    q = torch.randn(B, Lq, 288, device=dev) * 0.3
    k = torch.randn(B, Lk, 288, device=dev) * 0.3
    ones = torch.ones(B, Lk, 288, device=dev)
    # with V = 1 every output element equals sum_k P_drop = (kept probability mass)/(1-p)
    attention.dropout_state(dev).fill_(11)
    o1 = attention.attention_core(q, k, ones, None, 8, 0.1, 3)
    o1b = attention.attention_core(q, k, ones, None, 8, 0.1, 3)
    assert torch.equal(o1, o1b)                       # same counter + salt -> same mask
    assert abs(o1.mean().item() - 1.0) < 0.01         # E[keep/(1-p)] = 1
    assert o1.std().item() > 1e-3                     # but it IS random
    o2 = attention.attention_core(q, k, ones, None, 8, 0.1, 4)
    assert not torch.equal(o1, o2)                    # different call site
    attention.advance_dropout_state(dev)
    o3 = attention.attention_core(q, k, ones, None, 8, 0.1, 3)
    assert not torch.equal(o1, o3)                    # next step
    # backward regenerates the same mask: directional derivative check in fp32
    v = torch.randn(B, Lk, 288, device=dev)
    w = torch.randn(B, Lq, 288, device=dev)

    def f(qq, kk, vv):
        return (attention.attention_core(qq, kk, vv, None, 8, 0.1, 9) * w).sum()
    qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))
    f(qg, kg, vg).backward()
    for t, g in ((q, qg.grad), (k, kg.grad), (v, vg.grad)):
        d = torch.randn_like(t)
        eps = 1e-2
        args_p = [q, k, v]; args_m = [q, k, v]
        i = [id(x) for x in (q, k, v)].index(id(t))
        args_p[i] = t + eps * d; args_m[i] = t - eps * d
        num = (f(*args_p) - f(*args_m)).item() / (2 * eps)
        ana = (g * d).sum().item()
        assert abs(num - ana) <= 2e-2 * max(1.0, abs(ana)), (i, num, ana)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["self", "posself", "cross", "distinct"])
@pytest.mark.parametrize("batch_first", [True, False])
def test_module_on_gpu_matches_torch_multiheadattention(case, batch_first):
    """Output AND every gradient (inputs, packed in-projection weight/bias, out-projection) of the
    one-node in-projection + fused core against torch.nn.MultiheadAttention."""
    from eda_amd import attention
    torch.manual_seed(0)
    ref = torch.nn.MultiheadAttention(288, 8, dropout=0.1).eval().cuda()
    mine = attention.MultiheadAttention(288, 8, dropout=0.1).eval().cuda()
    with torch.no_grad():
        ref.in_proj_bias.normal_(0, 0.1); ref.out_proj.bias.normal_(0, 0.1)
    mine.load_state_dict(ref.state_dict())
    B, Lq, Lk = 4, 256, 80
    leaves = [torch.randn(B, Lq, 288, device="cuda", requires_grad=True),
              torch.randn(B, Lq, 288, device="cuda", requires_grad=True),
              torch.randn(B, Lk, 288, device="cuda", requires_grad=True),
              torch.randn(B, Lk, 288, device="cuda", requires_grad=True)]
    w = torch.randn(B, Lq, 288, device="cuda")

    def run(mod, is_ref):
        x, pos, mem, mem2 = leaves
        if case == "self":
            q = k = v = x; mask = _mask(B, Lq, 1).cuda()
        elif case == "posself":
            q = k = x + pos; v = x; mask = None
        elif case == "cross":
            q = x + pos; k = v = mem; mask = _mask(B, Lk, 2).cuda()
        else:
            q = x; k = mem; v = mem2; mask = _mask(B, Lk, 3).cuda()
        if is_ref or not batch_first:
            qt = q.transpose(0, 1)
            kt = qt if k is q else k.transpose(0, 1)
            vt = kt if v is k else (qt if v is q else v.transpose(0, 1))
            out = mod(qt, kt, vt, key_padding_mask=mask)[0].transpose(0, 1)
        else:
            out = mod(q, k, v, key_padding_mask=mask, batch_first=True)[0]
        params = [mod.in_proj_weight, mod.in_proj_bias, mod.out_proj.weight, mod.out_proj.bias]
        grads = torch.autograd.grad((out * w).sum(), leaves + params, allow_unused=True)
        return [out.detach()] + list(grads)

    exp = run(ref, True)
    got = run(mine, False)
    names = ["out", "dx", "dpos", "dmem", "dmem2", "dW_in", "db_in", "dW_out", "db_out"]
    for name, g, e in zip(names, got, exp):
        assert (g is None) == (e is None), name
        if e is None:
            continue
        scale = e.abs().max().item() + 1e-9
        assert (g - e).abs().max().item() <= 2e-4 * scale + 1e-6, (name, (g - e).abs().max().item(), scale)


# ------------------------------------------------------------------ 16-bit MFMA variants (csrc/mha2.hip, ArBf16 / ArF16)
# Tolerance (documented, BASELINE.json configs[2]/[4]): the contractions take bf16 / fp16 OPERANDS (8 / 11
# significant bits: unit roundoff 3.9e-3 / 4.9e-4) and accumulate in fp32; Q, K, V, P, dS and dO are each
# rounded once, so scores carry ~2u relative error, probabilities ~2u|s|, outputs / gradients a few u of the
# tensor's scale.  Bound used: 6u * max|expected| in max norm (bf16 2.4e-2, fp16 3e-3); the fp32 kernels
# (the parity path of the north star) keep the 1e-4 test above.
SHAPES16 = [(2, 80, 1024, False), (2, 1024, 80, True), (1, 1024, 1024, False), (2, 256, 132, True),
            (2, 130, 1024, False), (2, 256, 130, True), (3, 37, 101, True), (1, 5, 1, False), (2, 64, 65, True)]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [("bf16", 2.4e-2), ("f16", 3e-3)])
@pytest.mark.parametrize("B,Lq,Lk,masked", SHAPES16)
def test_16bit_mfma_attention_vs_restatement(B, Lq, Lk, masked, dtype, tol):
    from eda_amd import attention
    torch.manual_seed(Lq * 5 + Lk)
    dev = "cuda"
    q = torch.randn(B, Lq, 288, device=dev, requires_grad=True)
    k = torch.randn(B, Lk, 288, device=dev, requires_grad=True)
    v = torch.randn(B, Lk, 288, device=dev, requires_grad=True)
    mask = _mask(B, Lk, Lq + Lk).to(dev) if masked else None
    w = torch.randn(B, Lq, 288, device=dev)
    attention.set_compute_dtype(dtype)
    try:
        assert attention.compute_dtype() == dtype
        out = attention.attention_core(q, k, v, mask, 8, 0.0, 0)
        (out * w).sum().backward()
    finally:
        attention.set_compute_dtype("f32")
    got = [out.detach(), q.grad.clone(), k.grad.clone(), v.grad.clone()]
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    exp_out = attention_ref.attention_core(qd, kd, vd, mask, 8)
    (exp_out * w.double()).sum().backward()
    exp = [exp_out.detach(), qd.grad, kd.grad, vd.grad]
    # dq / dk are P * (dP - delta) contracted with K / Q: when the softmax is (nearly) one-hot the exact
    # value cancels to ~0 while dP (16-bit operands) and delta (fp32) each carry u * |dO||V| -- the error
    # is relative to the size of the cancelling terms, not of the result
    cancel = (w.abs().max() * v.abs().max() * 6.0).item()
    natural = {"out": 0.0, "dv": 0.0, "dq": cancel * k.abs().max().item() / 6.0, "dk": cancel * q.abs().max().item() / 6.0}
    for name, g, e in zip(["out", "dq", "dk", "dv"], got, exp):
        assert torch.isfinite(g).all(), name
        err = (g.double() - e).abs().max().item()
        scale = e.abs().max().item() + 1e-12
        assert err <= tol * max(scale, 0.05 * natural[name]), (name, err, scale)
        # and not trivially loose: the 16-bit result must differ from fp64 by more than fp32 rounding would
        # (guards against the dtype switch silently running the fp32 kernels)
        if e.numel() > 4096:
            assert err >= 1e-6 * scale, (name, "suspiciously exact for a 16-bit contraction", err)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_16bit_attention_dropout_consistency(dtype):
    """Keep rate of the attention-probability dropout and forward/backward mask consistency (V = 1 makes
    every output the kept probability mass / (1-p); d(loss)/dV then is the column sum of the dropped P)."""
    from eda_amd import attention
    dev = "cuda"
    torch.manual_seed(9)
    B, Lq, Lk, p = 2, 128, 192, 0.25
    q = torch.randn(B, Lq, 288, device=dev) * 0.3
    k = torch.randn(B, Lk, 288, device=dev) * 0.3
    ones = torch.ones(B, Lk, 288, device=dev, requires_grad=True)
    attention.dropout_state(dev).fill_(3)
    attention.set_compute_dtype(dtype)
    try:
        o1 = attention.attention_core(q, k, ones, None, 8, p, 77)
        o2 = attention.attention_core(q, k, ones, None, 8, p, 77)
        o3 = attention.attention_core(q, k, ones, None, 8, p, 78)
        o1.sum().backward()
    finally:
        attention.set_compute_dtype("f32")
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    mass = o1.detach()[..., ::36].mean().item()          # one column per head: E[kept mass / (1-p)] = 1
    assert abs(mass - 1.0) < 0.02, mass
    # sum_q out[q, d] = sum_k (sum_q Pd[q, k]) * 1  and  dV[k, d] = sum_q Pd[q, k]  ->  totals agree per head
    tot_out = o1.detach().view(B, Lq, 8, 36)[..., 0].sum(1)           # (B, 8)
    tot_dv = ones.grad.view(B, Lk, 8, 36)[..., 0].sum(1)
    torch.testing.assert_close(tot_out, tot_dv, rtol=2e-2, atol=1e-2)


# ---------------------------------------------- fused [q-projection | attention core] launch (short key sets) ----
@pytest.mark.gpu
@pytest.mark.parametrize("B,Lq,Lk,masked,p", [
    (8, 256, 80, True, 0.1), (8, 256, 132, True, 0.1), (2, 1024, 80, True, 0.0), (2, 1024, 132, True, 0.1),
    (2, 100, 130, True, 0.0), (1, 17, 1, False, 0.0), (3, 64, 192, False, 0.1), (2, 65, 17, True, 0.1), (2, 256, 64, False, 0.0)])
def test_qproj_fused_site_equals_projection_launch_plus_core_launch(B, Lq, Lk, masked, p, monkeypatch):
    """eda_mha_qproj_fwd (csrc/mha2.hip mha2_qproj_fwd_kernel: q = x Wq^T + bq and the attention core in ONE launch,
    the sites of encoder_decoder_layers.py:99-117, 375-391) against the two-launch composition it replaces, through the
    module: projected q, attention output, log-sum-exp and -- because the backward is shared and the Dropout hash is the
    same -- every gradient.  Then against the fp64 restatement directly (the q the kernel wrote, its output)."""
    from eda_amd import attention
    torch.manual_seed(Lq * 7 + Lk)
    dev = "cuda"
    mha = attention.MultiheadAttention(288, 8, dropout=p).to(dev).train()
    with torch.no_grad():
        mha.in_proj_bias.normal_(0, 0.2)
    x = torch.randn(B, Lq, 288, device=dev)
    mem = torch.randn(B, Lk, 288, device=dev)
    mask = _mask(B, Lk, 3).to(dev) if masked else None
    w = torch.randn(B, Lq, 288, device=dev)
    counter = attention.get_dropout_counter(torch.device(dev, 0))
    res = {}
    for fused in ("1", "0"):                  # (1 forces the fused launch for every supported shape; the default takes
        monkeypatch.setenv("EDA_MHA_QPROJ", fused)    # it only while it is one round of workgroups)
        attention.set_dropout_counter(torch.device(dev, 0), counter)
        xq = x.clone().requires_grad_(True)
        mm = mem.clone().requires_grad_(True)
        mha.zero_grad(set_to_none=True)
        timer_names = []
        from eda_amd import ext
        ext.op_timer = ext.OpTimer()
        try:
            o, _ = mha(xq, mm, mm, key_padding_mask=mask, batch_first=True, skip_out_proj=True)
            torch.cuda.synchronize()
            timer_names = [k[0] for k in ext.op_timer.summary()]
        finally:
            ext.op_timer = None
        (o * w).sum().backward()
        res[fused] = (o.detach(), xq.grad, mm.grad, mha.in_proj_weight.grad.clone(), mha.in_proj_bias.grad.clone(), timer_names)
    assert "mha_qproj_fwd" in res["1"][5] and "mha_fwd" not in res["1"][5]          # the fused launch really ran
    assert "mha_fwd" in res["0"][5] and "mha_qproj_fwd" not in res["0"][5]
    for a, b_, name in zip(res["1"][:5], res["0"][:5], ["out", "dx", "dmem", "dW", "db"]):
        scale = float(b_.abs().max()) + 1e-12
        assert float((a - b_).abs().max()) <= 2e-5 * scale, (name, float((a - b_).abs().max()), scale)
    if p == 0.0:
        # against the fp64 restatement (oracle/attention_ref.py) on the module's own parameters
        W, bb = mha.in_proj_weight.detach().double(), mha.in_proj_bias.detach().double()
        q64 = x.double() @ W[:288].t() + bb[:288]
        k64 = mem.double() @ W[288:576].t() + bb[288:576]
        v64 = mem.double() @ W[576:].t() + bb[576:]
        exp = attention_ref.attention_core(q64, k64, v64, mask, 8, 0.0, 0)
        got = res["1"][0].double()
        err = (got - exp).abs()
        assert float((err - 1e-4 * exp.abs()).max()) <= 4e-6 * float(exp.abs().max())
