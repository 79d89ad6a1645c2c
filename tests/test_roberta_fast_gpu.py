"""The frozen RoBERTa encoder on the repo's kernels (eda_amd/roberta_fast.py) against the Hugging Face
forward of the SAME module (the reference's text encoder: models/bdetr.py:77-80, 170-175): last_hidden_state within the
north star's 1e-4 for fp32 activations, for the padded lengths of BASELINE.json's configs (80 / 130 tokens), ragged
attention masks and a non-trivial number of layers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _encoder(layers, seed):
    from transformers import RobertaConfig, RobertaModel
    torch.manual_seed(seed)
    cfg = RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1,
                        num_hidden_layers=layers)
    m = RobertaModel(cfg).eval()
    with torch.no_grad():                       # LayerNorm affine / biases away from their 1 / 0 initialisation
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.05)
            elif "LayerNorm.weight" in n:
                p.uniform_(0.5, 1.5)
    return m


@pytest.mark.parametrize("L,layers", [(16, 1), (80, 3), (130, 2), (37, 1)])
def test_fast_encoder_equals_hugging_face_forward(L, layers):
    from eda_amd import roberta_fast, synthetic
    dev = "cuda"
    m = _encoder(layers, seed=L).to(dev)
    ids, am = synthetic.utterance_tokens(L, 4, max_len=L)
    ids, am = torch.from_numpy(ids).to(dev), torch.from_numpy(am).to(dev)
    with torch.no_grad():
        exp = m(input_ids=ids, attention_mask=am).last_hidden_state
    assert roberta_fast.supported(m, ids)
    got = roberta_fast.encode(m, ids, am)
    assert got.shape == exp.shape
    valid = am.bool()                            # (padded positions are computed by both but never read downstream)
    err = (got - exp).abs()[valid]
    tol = 1e-4 * exp.abs()[valid] + 1e-5 * exp.abs().max()
    assert (err <= tol).all(), (err.max().item(), exp.abs().max().item())
    # the stock module ran double-precision-free fp32 too: both are within rounding of an fp64 evaluation
    exp64 = m.double()(input_ids=ids, attention_mask=am).last_hidden_state
    e_fast = (got.double() - exp64).abs()[valid].max().item()
    e_hf = (exp.double() - exp64).abs()[valid].max().item()
    assert e_fast <= 4 * e_hf + 1e-5, (e_fast, e_hf)


def test_fast_encoder_follows_in_place_weight_updates_and_is_used_by_the_model():
    from eda_amd import roberta_fast, synthetic
    from eda_amd.bdetr import BeaUTyDETR
    dev = "cuda"
    model = BeaUTyDETR(num_queries=16, num_decoder_layers=1)
    model.text_encoder = _encoder(1, seed=3)
    model = model.to(dev).eval()
    ids, am = synthetic.utterance_tokens(1, 2, max_len=20)
    ids, am = torch.from_numpy(ids).to(dev), torch.from_numpy(am).to(dev)
    a = model.encode_text_frozen(ids, am)
    with torch.no_grad():
        exp = model.text_encoder(input_ids=ids, attention_mask=am).last_hidden_state
    torch.testing.assert_close(a, exp, rtol=1e-4, atol=1e-4)
    # load_state_dict writes the parameters in place: the packed q|k|v copies must be rebuilt
    sd = {k: (v * 1.5 if "attention.self.query.weight" in k else v) for k, v in model.text_encoder.state_dict().items()}
    model.text_encoder.load_state_dict(sd)
    b = model.encode_text_frozen(ids, am)
    with torch.no_grad():
        exp2 = model.text_encoder(input_ids=ids, attention_mask=am).last_hidden_state
    torch.testing.assert_close(b, exp2, rtol=1e-4, atol=1e-4)
    assert not torch.allclose(a, b)


def test_train_mode_runs_the_encoders_dropout_layers():
    """The reference trains with the whole model in train mode (main_utils.py:459): the frozen encoder's dropout layers are active.
    The fast forward then drops inside its own launches: p = 0 in the config equals the eval forward; with p = 0.1 two calls differ,
    the same counter value reproduces the same bits, outputs stay finite and near the eval forward on average."""
    import copy
    from eda_amd import roberta_fast
    from eda_amd import synthetic
    m = _encoder(2, seed=9).to("cuda")
    ids, am = synthetic.utterance_tokens(48, 6, max_len=48)
    ids, am = torch.from_numpy(ids).to("cuda"), torch.from_numpy(am).to("cuda")
    ref = roberta_fast.encode(m, ids, am)
    m0 = copy.deepcopy(m).train()
    m0.config.hidden_dropout_prob = 0.0
    m0.config.attention_probs_dropout_prob = 0.0
    assert roberta_fast.supported(m0, ids)
    assert torch.equal(roberta_fast.encode(m0, ids, am), ref)
    mt = copy.deepcopy(m).train()
    a = roberta_fast.encode(mt, ids, am)
    b = roberta_fast.encode(mt, ids, am)
    assert torch.isfinite(a).all() and not torch.equal(a, b)
    fast = mt.__dict__["_eda_fast"]
    fast.counter.sub_(2)                                   # back to the counter of the first call
    assert torch.equal(roberta_fast.encode(mt, ids, am), a)
    keep = am.bool()[..., None].expand_as(a)
    mean = torch.stack([roberta_fast.encode(mt, ids, am) for _ in range(24)]).mean(0)
    d_mean = float((mean - ref)[keep].abs().mean())
    d_one = float((a - ref)[keep].abs().mean())
    assert d_one > 1e-3 and d_mean < 0.6 * d_one, (d_one, d_mean)
    # an in-place weight update makes the packed copy stale: the repacked forward continues the same mask stream
    c0 = roberta_fast.get_dropout_counter(mt)
    with torch.no_grad():
        mt.embeddings.LayerNorm.bias.add_(0.0)
    roberta_fast.encode(mt, ids, am)
    assert mt.__dict__["_eda_fast"] is not fast and roberta_fast.get_dropout_counter(mt) == c0 + 1
    roberta_fast.set_dropout_counter(mt, c0 - 24 - 1)      # (checkpoint resume: eda_amd/checkpoint.py)
    assert torch.equal(roberta_fast.encode(mt, ids, am), a)


def test_dropout_primitives():
    from eda_amd import _lib
    from eda_amd.roberta_fast import attention_hd64
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    cnt = torch.full((1,), 1234567, dtype=torch.int64, device=dev)
    x = torch.randn(1000003 // 4 * 4 + 3, device=dev).abs() + 0.5
    out = torch.empty_like(x)
    _lib.check(_lib.lib().eda_dropout_f32(x.data_ptr(), x.numel(), 0.1, cnt.data_ptr(), 7, out.data_ptr(), s), "eda_dropout_f32")
    kept = out != 0
    assert abs(float(kept.float().mean()) - 0.9) < 2e-3
    torch.testing.assert_close(out[kept], x[kept] / 0.9, rtol=1e-6, atol=0)
    out2 = torch.empty_like(x)
    _lib.check(_lib.lib().eda_dropout_f32(x.data_ptr(), x.numel(), 0.1, cnt.data_ptr(), 8, out2.data_ptr(), s), "eda_dropout_f32")
    assert not torch.equal(out2 != 0, kept)                # another call site, another mask
    # attention: p = 0 is the plain entry; with v = 1 every output is sum_k keep_k p_k / (1 - p): mean 1, and it varies
    B, L, H = 4, 96, 12
    q, k = torch.randn(B, L, 64 * H, device=dev), torch.randn(B, L, 64 * H, device=dev)
    v = torch.ones(B, L, 64 * H, device=dev)
    kpm = torch.zeros(B, L, dtype=torch.bool, device=dev)
    kpm[1, 70:] = True
    o0 = attention_hd64(q, k, v, kpm, H, 0.125)
    torch.testing.assert_close(o0, torch.ones_like(o0), rtol=1e-5, atol=1e-5)
    o = attention_hd64(q, k, v, kpm, H, 0.125, 0.1, cnt, 3)
    assert abs(float(o.mean()) - 1.0) < 5e-3 and float(o.std()) > 0.01
    cols = o.view(B, L, H, 64)
    assert float((cols - cols[..., :1]).abs().max()) < 1e-5          # one mask per (query, key): every output dim sees the same one
