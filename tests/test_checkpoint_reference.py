"""Checkpoint interchange with the REFERENCE (main_utils.py:126-166, :279-301), exercised with the
reference's own model and optimizer classes imported in the build container (tools/ref_import.py) -- skipped where
/root/reference does not exist (the GPU box):
  * a checkpoint written in the reference's layout from the reference model (DistributedDataParallel `module.` prefix,
    config namespace, per-parameter AdamW state of the three learning-rate groups) loads strictly into
    eda_amd.BeaUTyDETR and -- through FlatParams' layout -- into the flat optimizer: the next update step equals the
    reference's;
  * a checkpoint written by eda_amd.checkpoint.save_checkpoint loads into the reference model and optimizer (strict)
    and continues identically."""
import argparse
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="needs the reference checkout (build container only)")

LRS = {"base": 1e-4, "backbone_net": 1e-3, "text_encoder": 1e-5}           # scripts/train_scanrefer.sh


def _ref_optimizer(model):
    named = list(model.named_parameters())                                  # main_utils.py:279-301
    groups = [{"params": [p for n, p in named if "backbone_net" not in n and "text_encoder" not in n and p.requires_grad]},
              {"params": [p for n, p in named if "backbone_net" in n and p.requires_grad], "lr": LRS["backbone_net"]},
              {"params": [p for n, p in named if "text_encoder" in n and p.requires_grad], "lr": LRS["text_encoder"]}]
    return torch.optim.AdamW(groups, lr=LRS["base"], weight_decay=5e-4)


def _grad(name, shape, seed):
    import model_fixtures as MF
    return MF.det_tensor(name + ".grad", shape, seed=seed) * 0.01


def _step_ref(model, opt, seed):
    for n, p in model.named_parameters():
        if p.requires_grad:
            p.grad = _grad(n, p.shape, seed)
    opt.step()


def _step_flat(model, flat, opt, seed):
    flat.flat_grad.zero_()
    shapes = {n: p.shape for n, p in model.named_parameters()}
    for n, _, off, numel in flat.layout:
        flat.flat_grad[off:off + numel].copy_(_grad(n, shapes[n], seed).reshape(-1))
    opt.step()


def test_reference_checkpoint_interchange(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_import
    import model_fixtures as MF
    from eda_amd import checkpoint
    from eda_amd.bdetr import BeaUTyDETR
    from eda_amd.parallel import FlatParams, reference_lr_groups
    mods = ref_import.load()
    ref = ref_import.build_reference_model(mods, seed=0, num_queries=64, butd=True)
    ref.text_encoder = MF.small_roberta(1)
    for p in ref.text_encoder.parameters():
        p.requires_grad = False                                             # bdetr.py:78-80
    MF.fill_det_state(ref, seed=30)
    ref_opt = _ref_optimizer(ref)
    assert len(ref_opt.param_groups) == 3 and len(ref_opt.param_groups[2]["params"]) == 0
    for seed in (77, 78):
        _step_ref(ref, ref_opt, seed)
    # ---- what the reference's save_checkpoint writes (main_utils.py:149-166), from the DDP wrapper's state_dict
    path = str(tmp_path / "ckpt_epoch_7.pth")
    torch.save({"config": argparse.Namespace(lr=1e-4, lr_backbone=1e-3, text_encoder_lr=1e-5, weight_decay=5e-4),
                "save_path": path, "model": {"module." + k: v for k, v in ref.state_dict().items()},
                "optimizer": ref_opt.state_dict(), "scheduler": {}, "epoch": 7}, path)

    torch.manual_seed(123)
    mine = BeaUTyDETR(num_queries=64, butd=True)
    mine.text_encoder = MF.small_roberta(1)
    for p in mine.text_encoder.parameters():
        p.requires_grad = False
    flat = FlatParams(mine, reference_lr_groups)
    opt = torch.optim.AdamW([{"params": [gp], "lr": LRS[k]} for k, gp in flat.groups.items()], weight_decay=5e-4)
    epoch, missing, unexpected = checkpoint.load_checkpoint(mine, path, optimizer=opt, flat=flat)
    assert epoch == 7 and not [k for k in missing + unexpected if "position_ids" not in k and "token_type_ids" not in k]
    sd_ref, sd_mine = ref.state_dict(), mine.state_dict()
    for k, v in sd_ref.items():
        if k in sd_mine:
            assert torch.equal(sd_mine[k], v), k
    # the parameters still live in the flat buffer
    assert all(p.data_ptr() >= flat.flat_param.data_ptr() for p in flat.params)
    # ---- the next update: reference AdamW on per-parameter state vs flat AdamW on the converted state
    _step_ref(ref, ref_opt, 79)
    _step_flat(mine, flat, opt, 79)
    pr = dict(ref.named_parameters())
    worst = max(((p.detach() - pr[n].detach()).abs().max() / (pr[n].detach().abs().max() + 1e-12)).item()
                for n, p in mine.named_parameters() if p.requires_grad)
    assert worst < 1e-6, worst

    # ---- and back: written here, loaded by the reference's classes (strict, as main_utils.py:135 does)
    path2 = str(tmp_path / "ckpt_epoch_8.pth")
    checkpoint.save_checkpoint(mine, path2, optimizer=opt, epoch=8, flat=flat,
                               config=argparse.Namespace(lr=1e-4))
    ck = torch.load(path2, map_location="cpu", weights_only=False)
    assert set(ck) >= {"config", "save_path", "model", "optimizer", "scheduler", "epoch"} and ck["epoch"] == 8
    ref2 = ref_import.build_reference_model(mods, seed=1, num_queries=64, butd=True)
    ref2.text_encoder = MF.small_roberta(1)
    for p in ref2.text_encoder.parameters():
        p.requires_grad = False
    hf = ("text_encoder.embeddings.position_ids", "text_encoder.embeddings.token_type_ids")
    sd = {k[len("module."):]: v for k, v in ck["model"].items()}
    res = ref2.load_state_dict(sd, strict=False)
    assert not [k for k in res.missing_keys + res.unexpected_keys if k not in hf]
    ref_opt2 = _ref_optimizer(ref2)
    ref_opt2.load_state_dict(ck["optimizer"])
    _step_ref(ref2, ref_opt2, 80)
    _step_flat(mine, flat, opt, 80)
    pr2 = dict(ref2.named_parameters())
    worst = max(((p.detach() - pr2[n].detach()).abs().max() / (pr2[n].detach().abs().max() + 1e-12)).item()
                for n, p in mine.named_parameters() if p.requires_grad)
    assert worst < 1e-6, worst
