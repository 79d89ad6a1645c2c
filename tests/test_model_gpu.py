"""The same model-layer parity cases on the product path: HIP kernels through the
C ABI on an MI355X, against goldens from the reference's Python layers."""
import pytest

import model_cases as MC

pytestmark = pytest.mark.gpu


def test_query_and_group():
    MC.run_query_and_group("cuda")


def test_sa_and_fp_modules():
    MC.run_fp_module("cuda")


def test_sa_module_train_mode():
    MC.run_sa_module_train("cuda")


def test_backbone():
    MC.run_backbone("cuda")


@pytest.mark.parametrize("butd", [True, False])
def test_encoder_decoder(butd):
    MC.run_encoder_decoder("cuda", butd)


@pytest.mark.parametrize("butd", [True, False])
def test_full_model(butd):
    MC.run_full_model("cuda", butd)


def test_full_model_130_tokens():
    """SR3D-shaped long utterances (BASELINE.json configs[4]): padded length 130, fp32 attention."""
    MC.run_full_model("cuda", True, tag="butd_l130")


# max |error| / (u * max|golden|) allowed with 16-bit attention (u = 2^-8 bf16, 2^-11 f16; derivation in
# model_cases.run_full_model).  Measured on MI355X (round 3): bf16 -- seed objectness logits 7.9 u (butd) / 9.6 u
# (no butd), every other tensor <= 1.7 u, 2 of the 128 selected queries differ from the fp32 golden's; f16 at 130
# tokens -- logits 3.1 u, the rest <= 0.8 u, identical query set.  The objectness logits sit behind the fixture's x40
# scaling of the last objectness layer (it widens the top-k gaps for the fp32 tests), hence their own bound.
K_16BIT = {"seeds_obj_cls_logits": 20.0, "default": 4.0}
MAX_QUERIES_CHANGED = 8          # of 128 (measured 0-5 across builds / dtypes); decided and documented: the top-k is NOT kept in fp32 (model_cases.run_full_model)


@pytest.mark.parametrize("dtype,tag", [("bf16", "butd"), ("bf16", "nobutd"), ("f16", "butd_l130")])
def test_full_model_16bit_attention(dtype, tag):
    """BASELINE.json configs[2] (bf16) and configs[4] (fp16, 130 tokens) at MODEL level: the whole forward with 16-bit
    MFMA attention against the reference's fp32 golden, incl. the stability of the selected query set under the
    16-bit noise (per-query tensors are compared on the queries both runs selected)."""
    rep = MC.run_full_model("cuda", tag != "nobutd", tag=tag, attn_dtype=dtype)
    top = sorted(rep["worst"].items(), key=lambda kv: -kv[1])[:5]
    print("16-bit attention report", dtype, tag, {k: (round(v, 2) if isinstance(v, float) else v)
                                                    for k, v in rep.items() if k != "worst"})
    print("  largest errors in units of u * max|golden|:", [(k, round(v, 2)) for k, v in top])
    assert rep["queries_changed"] <= MAX_QUERIES_CHANGED, rep["queries_changed"]
    for k, use in rep["worst"].items():
        assert use <= K_16BIT.get(k, K_16BIT["default"]), (k, use, top)


def test_hoisted_decoder_kv_projections_equal_per_layer_projections(monkeypatch):
    """The decoder's K | V projections of text / boxes / points for all six layers as one product each
    (BeaUTyDETR._hoisted_kv) against the per-layer projections, in TRAIN mode with Dropout (same counter, same masks): same
    outputs, same parameter gradients up to fp32 rounding (the memories' input gradients are one long contraction instead
    of six products and five additions)."""
    import torch
    from eda_amd import attention
    from eda_amd.bdetr import BeaUTyDETR
    from tests import model_fixtures as MF
    dev = "cuda"
    torch.manual_seed(3)
    model = BeaUTyDETR(num_queries=64, butd=True)
    model.text_encoder = MF.small_roberta(1)
    MF.fill_det_state(model, seed=30)
    model.train().to(dev)
    inputs = MF.full_model_inputs(7, max_len=16)
    inputs = {k: ({kk: vv.to(dev) for kk, vv in v.items()} if isinstance(v, dict) else v.to(dev)) for k, v in inputs.items()}
    from eda_amd import roberta_fast
    bn_state = {k: v.clone() for k, v in model.state_dict().items()}
    counter = attention.get_dropout_counter(torch.device(dev))
    res = {}
    for hoist in ("1", "0"):
        monkeypatch.setenv("EDA_HOIST_KV", hoist)
        model.load_state_dict(bn_state)                      # (running statistics move in train mode)
        attention.set_dropout_counter(torch.device(dev), counter)
        roberta_fast.set_dropout_counter(model.text_encoder, 777)      # (the frozen encoder's own mask stream, train mode)
        torch.manual_seed(11)                                # (anything that draws from torch's generator)
        model.zero_grad(set_to_none=True)
        ep = model(inputs)
        assert bool(model._hoisted_kv.__func__) and (hoist == "0" or getattr(model, "_kv_stacks", None))
        keys = sorted(k for k, v in ep.items() if torch.is_tensor(v) and v.dtype.is_floating_point and v.requires_grad)
        loss = sum((ep[k].float() ** 2).mean() for k in keys)
        loss.backward()
        res[hoist] = ({k: ep[k].detach().clone() for k in keys},
                      {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    (o1, g1), (o0, g0) = res["1"], res["0"]
    assert set(o1) == set(o0) and set(g1) == set(g0) and len(g1) > 300
    for k in o0:
        assert float((o1[k] - o0[k]).abs().max()) <= 2e-5 * (float(o0[k].abs().max()) + 1e-12), k
    # (biases in front of a train-mode BatchNorm have a zero gradient in exact arithmetic: noise of the size of the
    # rounding errors in both runs -- hence the absolute term, relative to the largest gradient of the model)
    gmax = max(float(v.abs().max()) for v in g0.values())
    for k in g0:
        scale = float(g0[k].abs().max())
        assert float((g1[k] - g0[k]).abs().max()) <= 2e-3 * scale + 1e-6 * gmax, (k, float((g1[k] - g0[k]).abs().max()), scale, gmax)
