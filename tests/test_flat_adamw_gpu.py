"""csrc/optim.hip (eda_amd.parallel.FlatClipAdamW): clip_grad_norm_ + AdamW of a flat-parameter model as two launches, against
FlatParams.clip_grad_norm_ + torch.optim.AdamW(fused, capturable) -- the reference's main_utils.py:277-305, 483-486 -- on the same
gradients: parameters and optimizer state after several steps, a learning-rate change between steps, the all-reduce pre-scale,
and replays of a captured graph that follow a scheduler through the pinned learning-rate array."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model():
    torch.manual_seed(0)
    m = torch.nn.ModuleDict({
        "backbone_net": torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 11)),      # odd sizes: groups end inside granules
        "head": torch.nn.Sequential(torch.nn.Linear(11, 129), torch.nn.LayerNorm(129), torch.nn.Linear(129, 3)),
    }).cuda()
    return m


def _setup():
    from eda_amd.parallel import FlatParams, reference_lr_groups
    m = _model()
    flat = FlatParams(m, reference_lr_groups)
    lrs = {"base": 1e-3, "backbone_net": 1e-2}
    opt = torch.optim.AdamW([{"params": [gp], "lr": lrs[k]} for k, gp in flat.groups.items()], weight_decay=5e-4, fused=True,
                            capturable=True)
    return m, flat, opt


def _grads(flat, seed, scale):
    g = torch.Generator(device="cuda").manual_seed(seed)
    flat.flat_grad.copy_(torch.randn(flat.flat_grad.shape, device="cuda", generator=g) * scale)
    # (padding floats between parameters carry zero gradients)
    mask = torch.zeros_like(flat.flat_grad)
    for v in flat._grad_views:
        v.fill_(1.0)
    mask.copy_(flat.flat_grad)
    flat.flat_grad.copy_(torch.randn(flat.flat_grad.shape, device="cuda", generator=g) * scale * mask)


@pytest.mark.parametrize("pre_scale,max_norm", [(1.0, 0.1), (0.125, 0.1), (1.0, 1e9), (1.0, 0.0)])
def test_two_launches_equal_clip_plus_torch_adamw(pre_scale, max_norm):
    from eda_amd.parallel import FlatClipAdamW
    _, fa, oa = _setup()
    _, fb, ob = _setup()
    fused = FlatClipAdamW(fb, ob)
    for step in range(6):
        scale = [1.0, 1e-3, 30.0][step % 3]
        _grads(fa, 10 + step, scale)
        fb.flat_grad.copy_(fa.flat_grad)
        if step == 3:                                   # a scheduler step
            for o in (oa, ob):
                for g in o.param_groups:
                    g["lr"] *= 0.1
        if max_norm > 0:
            na = fa.clip_grad_norm_(max_norm, pre_scale=pre_scale)
        else:
            na = torch.linalg.vector_norm(fa.flat_grad) * pre_scale
            fa.flat_grad.mul_(pre_scale)
        oa.step()
        nb = fused.step(max_norm, pre_scale)
        torch.testing.assert_close(nb.reshape(()), na.reshape(()), rtol=2e-6, atol=0)
        torch.testing.assert_close(fb.flat_param, fa.flat_param, rtol=2e-6, atol=2e-8)      # (an ulp of a parameter of size 0.1)
    for ga, gb in zip(oa.param_groups, ob.param_groups):
        sa, sb = oa.state[ga["params"][0]], ob.state[gb["params"][0]]
        assert float(sa["step"]) == float(sb["step"]) == 6.0
        # (differences: the clip coefficient's last bit -- the norm is summed in another order -- through a lerp of mixed signs)
        torch.testing.assert_close(sb["exp_avg"], sa["exp_avg"], rtol=2e-6, atol=2e-6 * float(sa["exp_avg"].abs().max()))
        torch.testing.assert_close(sb["exp_avg_sq"], sa["exp_avg_sq"], rtol=4e-6, atol=4e-6 * float(sa["exp_avg_sq"].abs().max()))
    # the optimizer object stays the owner of the state: its state_dict round-trips into a fresh torch AdamW
    _, fc, oc = _setup()
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))
    assert float(oc.state[oc.param_groups[0]["params"][0]]["step"]) == 6.0


def test_captured_step_follows_the_learning_rates_and_is_reproducible():
    from eda_amd.parallel import FlatClipAdamW
    _, fa, oa = _setup()
    _, fb, ob = _setup()
    eager, captured = FlatClipAdamW(fa, oa), FlatClipAdamW(fb, ob)
    _grads(fa, 5, 1.0)
    fb.flat_grad.copy_(fa.flat_grad)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        captured.step(0.1, 1.0)                         # warm-up (also step 1 of the comparison)
        eager.step(0.1, 1.0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            captured.step(0.1, 1.0)
        # the capture itself ran nothing: replay is step 2
        for it in range(4):
            if it == 2:
                for o in (oa, ob):
                    for grp in o.param_groups:
                        grp["lr"] *= 0.5
            captured.sync_lrs()                         # host write into the pinned array the graph reads
            g.replay()
            eager.step(0.1, 1.0)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    assert torch.equal(fb.flat_param, fa.flat_param)     # same kernels, same order: bit-identical
    assert float(ob.state[ob.param_groups[0]["params"][0]]["step"]) == 5.0
