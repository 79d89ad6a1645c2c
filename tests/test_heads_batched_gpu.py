"""The seven prediction heads' backward passes issued together (eda_amd/heads_batched.py) against the per-head autograd
nodes (EDA_BATCHED_HEADS=0): same forward numbers bit for bit, every gradient to fp32 rounding (the heads' first-layer input
gradient is a grouped launch there, a W^T-form product here)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(batched, defer, what="EDA_BATCHED_HEADS"):
    from eda_amd import synthetic
    from eda_amd.bdetr import BeaUTyDETR
    from eda_amd.parallel import FlatParams
    import itertools
    from eda_amd import attention, fused_ln
    os.environ[what] = "1" if batched else "0"
    try:
        # the same dropout streams in both runs: call-site salts are drawn from process-wide counters at construction
        attention._salt_counter = itertools.count(1)
        fused_ln._salt_counter = itertools.count(1 << 20)
        torch.manual_seed(0)
        dev = torch.device("cuda", 0)
        model = BeaUTyDETR(num_queries=64, num_decoder_layers=2).to(dev).train()
        flat = FlatParams(model)
        pc = synthetic.batch([0, 1], 6000)
        ids, am = synthetic.utterance_tokens(0, 2, max_len=12)
        boxes, bmask, cls = synthetic.detected_boxes(0, 2)
        inputs = {"point_clouds": torch.from_numpy(pc).to(dev),
                  "tokenized": {"input_ids": torch.from_numpy(ids).to(dev), "attention_mask": torch.from_numpy(am).to(dev)},
                  "det_boxes": torch.from_numpy(boxes).to(dev), "det_bbox_label_mask": torch.from_numpy(bmask).to(dev),
                  "det_class_ids": torch.from_numpy(cls).to(dev)}
        attention.set_dropout_counter(dev, 1234)
        ep = model(inputs)
        keys = sorted(k for k in ep if any(k.endswith(s) for s in ("center", "pred_size", "sem_cls_scores")))
        loss = sum((ep[k] * torch.linspace(0.5, 1.5, ep[k].numel(), device=dev).view_as(ep[k])).sum() for k in keys)
        loss = loss + ep["last_proj_queries"].pow(2).sum() if "last_proj_queries" in ep else loss
        if defer:
            with flat.deferred_wgrad():
                loss.backward()
            flat.collect_grads()
        else:
            loss.backward()
            flat.collect_grads()
        return {k: ep[k].detach().clone() for k in keys}, flat.flat_grad.clone(), float(loss)
    finally:
        os.environ.pop(what, None)


@pytest.mark.parametrize("defer", [False, True])
def test_batched_heads_backward_equals_per_head_nodes(defer):
    out_b, g_b, l_b = _run(True, defer)
    out_p, g_p, l_p = _run(False, defer)
    assert out_b.keys() == out_p.keys() and len(out_b) >= 9
    for k in out_b:
        assert torch.equal(out_b[k], out_p[k]), k           # the forward launches are the same ones
    assert l_b == l_p
    assert torch.isfinite(g_b).all() and g_b.abs().max() > 0
    scale = g_p.abs().max().item()
    assert (g_b - g_p).abs().max().item() <= 2e-5 * scale
    # the heads' own parameters saw gradients on the batched path
    nz = (g_b != 0).float().mean().item()
    assert abs(nz - (g_p != 0).float().mean().item()) < 1e-3


@pytest.mark.parametrize("defer", [False, True])
def test_batched_posembed_backward_equals_per_layer_nodes(defer):
    """The decoder layers' positional embeddings: backward at the end of the pass, all layers together
    (eda_amd/posembed_batched.py), against the per-layer autograd nodes."""
    out_b, g_b, l_b = _run(True, defer, "EDA_BATCHED_POSEMBED")
    out_p, g_p, l_p = _run(False, defer, "EDA_BATCHED_POSEMBED")
    for k in out_b:
        torch.testing.assert_close(out_b[k], out_p[k], rtol=1e-5, atol=1e-6)
    scale = g_p.abs().max().item()
    assert torch.isfinite(g_b).all()
    assert (g_b - g_p).abs().max().item() <= 2e-5 * scale
    assert abs((g_b != 0).float().mean().item() - (g_p != 0).float().mean().item()) < 1e-3


def test_batched_posembed_is_opt_in_and_autograd_delivers_without_it():
    """the batched form assigns `.grad` from an end-of-backward callback, so hooks keyed on autograd delivery
    (DDP's reducer) would never see these gradients.  Without FlatParams (or with a tensor hook on one of the six
    parameters) the per-layer autograd nodes run and every hook fires."""
    from eda_amd import posembed_batched, synthetic
    from eda_amd.bdetr import BeaUTyDETR
    from eda_amd.parallel import FlatParams
    was = posembed_batched.enabled()
    try:
        posembed_batched.enable(False)
        torch.manual_seed(0)
        dev = torch.device("cuda", 0)
        model = BeaUTyDETR(num_queries=64, num_decoder_layers=2).to(dev).train()
        head = model.decoder[0].self_posembed.position_embedding_head
        six = [head[0].weight, head[0].bias, head[1].weight, head[1].bias, head[3].weight, head[3].bias]
        seen = []
        handles = [p.register_hook(lambda g, i=i: seen.append(i)) for i, p in enumerate(six)]
        pc = synthetic.batch([0, 1], 6000)
        ids, am = synthetic.utterance_tokens(0, 2, max_len=12)
        boxes, bmask, cls = synthetic.detected_boxes(0, 2)
        inputs = {"point_clouds": torch.from_numpy(pc).to(dev),
                  "tokenized": {"input_ids": torch.from_numpy(ids).to(dev), "attention_mask": torch.from_numpy(am).to(dev)},
                  "det_boxes": torch.from_numpy(boxes).to(dev), "det_bbox_label_mask": torch.from_numpy(bmask).to(dev),
                  "det_class_ids": torch.from_numpy(cls).to(dev)}
        ep = model(inputs)
        (ep["last_center"].pow(2).sum() + ep["last_sem_cls_scores"].pow(2).mean()).backward()
        assert sorted(set(seen)) == list(range(6)), seen
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in six)
        for h in handles:
            h.remove()
        # the owner that reads `.grad` itself switches the batched form on ...
        FlatParams(model)
        assert posembed_batched.enabled()
        probe = torch.zeros(2, 64, 6, device=dev)
        assert posembed_batched.PosEmbedBatch.usable(model.decoder[0].self_posembed, probe)
        # ... and a hooked parameter still selects the autograd nodes
        h = six[0].register_hook(lambda g: None)
        assert not posembed_batched.PosEmbedBatch.usable(model.decoder[0].self_posembed, probe)
        h.remove()
    finally:
        posembed_batched.enable(was)
