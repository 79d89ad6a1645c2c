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
