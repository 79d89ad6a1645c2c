"""bench.synthetic_loss (SURVEY §8d "Loss for bwd") has three evaluation orders -- per head (the definition), heads stacked
(rounds 1-5), one weighted dot product over all squared terms (round 6) -- that must be the same function: value and every
gradient to fp32 rounding."""
import torch


def test_the_three_evaluation_orders_of_the_synthetic_loss_agree(monkeypatch):
    import bench
    torch.manual_seed(0)
    B, Q, L = 2, 8, 5
    ep = {"seeds_obj_cls_logits": torch.randn(B, 1, 16, requires_grad=True), "proj_tokens": torch.randn(B, L, 6, requires_grad=True)}
    for p in ["proposal_", "0head_", "1head_", "last_"]:
        ep[p + "center"] = torch.randn(B, Q, 3, requires_grad=True)
        ep[p + "pred_size"] = torch.randn(B, Q, 3, requires_grad=True)
        ep[p + "sem_cls_scores"] = torch.randn(B, Q, 7, requires_grad=True)
        ep[p + "proj_queries"] = torch.randn(B, Q, 6, requires_grad=True)
    res = {}
    for form in ("perhead", "stacked", "flat"):
        monkeypatch.setenv("EDA_BENCH_LOSS_FORM", form)
        for t in ep.values():
            t.grad = None
        loss = bench.synthetic_loss(ep)
        loss.backward()
        res[form] = (float(loss.detach()), {k: v.grad.clone() for k, v in ep.items()})
    ref_l, ref_g = res["perhead"]
    for form in ("stacked", "flat"):
        l, g = res[form]
        assert abs(l - ref_l) <= 1e-5 * abs(ref_l), (form, l, ref_l)
        for k in ep:
            assert (g[k] - ref_g[k]).abs().max().item() <= 1e-6 * (ref_g[k].abs().max().item() + 1e-12), (form, k)
