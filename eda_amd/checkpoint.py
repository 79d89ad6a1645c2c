"""Checkpoint I/O in the reference's format (main_utils.py:126-166).

The reference saves ``{'config', 'save_path', 'model', 'optimizer', 'scheduler', 'epoch'}`` where
``model`` is the state_dict of the DistributedDataParallel WRAPPER (every key carries a ``module.``
prefix, main_utils.py:155) and loads it back strictly into the wrapper (:135).  The modules of this
repo keep the reference's parameter / buffer names (tests/golden/state_dict_manifest.json, 805
non-RoBERTa tensors), so a reference checkpoint loads into ``eda_amd.bdetr.BeaUTyDETR`` and a
checkpoint written here loads into the reference:

* ``load_checkpoint(model, path)``   strips / tolerates the ``module.`` prefix, loads strictly, except for
  Hugging Face bookkeeping buffers that exist in one transformers version and not in the other
  (``text_encoder.embeddings.position_ids`` / ``token_type_ids``: environment.yml pins 4.17, this image
  has 5.x);
* ``save_checkpoint(...)``           writes the same dictionary layout with the ``module.`` prefix.

With ``eda_amd.parallel.FlatParams`` the parameters are views of one flat buffer; ``load_state_dict``
copies INTO those views, so loading after FlatParams construction keeps the flat layout intact.
"""
import argparse
import os

import torch

_HF_BOOKKEEPING = ("text_encoder.embeddings.position_ids", "text_encoder.embeddings.token_type_ids")


def _strip(sd):
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def _read(path, map_location, trusted):
    """torch.load restricted to tensors / containers (+ the argparse.Namespace the reference stores under 'config');
    arbitrary pickles only with trusted=True."""
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location=map_location, weights_only=True)
    except Exception:
        if not trusted:
            raise
        return torch.load(path, map_location=map_location, weights_only=False)


def reference_param_order(model):
    """The parameter order of the reference's AdamW (main_utils.py:279-301): three groups -- neither backbone nor text
    encoder | backbone_net | text_encoder -- each in named_parameters() order, trainable parameters only."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    groups = [[n for n, _ in named if "backbone_net" not in n and "text_encoder" not in n],
              [n for n, _ in named if "backbone_net" in n],
              [n for n, _ in named if "text_encoder" in n]]
    return groups


def reference_optimizer_state_to_flat(ref_state, model, flat, optimizer):
    """Load the state_dict of the reference's per-parameter AdamW into the flat optimizer (one tensor per learning-rate
    group, eda_amd.parallel.FlatParams): exp_avg / exp_avg_sq of every parameter are copied into its slice of the
    group's flat moment tensors (padding stays zero), `step` is taken from the parameters (they all agree)."""
    groups = reference_param_order(model)
    pg = ref_state["param_groups"]
    where = {n: (k, off, numel) for n, k, off, numel in flat.layout}
    flat_params = {k: gp for k, gp in flat.groups.items()}
    new_state = {}
    for gi, names in enumerate(groups):
        if gi >= len(pg):
            break
        ids = pg[gi]["params"]
        if len(ids) != len(names):
            raise RuntimeError(f"optimizer group {gi}: checkpoint has {len(ids)} parameters, the model {len(names)}")
        for pid, name in zip(ids, names):
            st = ref_state["state"].get(pid)
            if st is None:
                continue
            k, off, numel = where[name]
            lo = flat.group_bounds[k][0]
            gp = flat_params[k]
            ns = new_state.setdefault(k, {"step": st["step"], "exp_avg": torch.zeros_like(gp.data),
                                          "exp_avg_sq": torch.zeros_like(gp.data)})
            ns["exp_avg"][off - lo:off - lo + numel].copy_(st["exp_avg"].reshape(-1))
            ns["exp_avg_sq"][off - lo:off - lo + numel].copy_(st["exp_avg_sq"].reshape(-1))
    for group in optimizer.param_groups:
        for gp in group["params"]:
            key = next(k for k, v in flat_params.items() if v is gp)
            if key in new_state:
                st = optimizer.state[gp]
                ns = new_state[key]
                step = ns["step"]
                st["step"] = (step.clone().to(gp.device).float() if torch.is_tensor(step) else
                              torch.tensor(float(step), dtype=torch.float32, device=gp.device if group.get("capturable") or group.get("fused") else "cpu"))
                st["exp_avg"], st["exp_avg_sq"] = ns["exp_avg"], ns["exp_avg_sq"]


def flat_optimizer_state_to_reference(model, flat, optimizer):
    """The inverse: a state_dict the reference's AdamW.load_state_dict accepts (hyper-parameters of the param_groups are
    copied from the flat optimizer's groups in the reference's order rest | backbone_net | text_encoder)."""
    groups = reference_param_order(model)
    where = {n: (k, off, numel) for n, k, off, numel in flat.layout}
    shapes = {n: p.shape for n, p in model.named_parameters()}
    by_key = {}
    for group in optimizer.param_groups:
        for gp in group["params"]:
            key = next(k for k, v in flat.groups.items() if v is gp)
            by_key[key] = (group, optimizer.state.get(gp, {}))
    state, param_groups, pid = {}, [], 0
    for names in groups:
        if not names:          # (the frozen text encoder: the reference still carries its -- empty -- group)
            g = {k: v for k, v in optimizer.param_groups[0].items() if k != "params"}
            g["params"] = []
            param_groups.append(g)
            continue
        key = where[names[0]][0]
        group, st = by_key[key]
        lo = flat.group_bounds[key][0]
        ids = []
        for n in names:
            _, off, numel = where[n]
            if st:
                step = st["step"]
                state[pid] = {"step": step.detach().clone().cpu() if torch.is_tensor(step) else torch.tensor(float(step)),
                              "exp_avg": st["exp_avg"][off - lo:off - lo + numel].detach().clone().view(shapes[n]).cpu(),
                              "exp_avg_sq": st["exp_avg_sq"][off - lo:off - lo + numel].detach().clone().view(shapes[n]).cpu()}
            ids.append(pid)
            pid += 1
        g = {k: v for k, v in group.items() if k != "params"}
        g["params"] = ids
        param_groups.append(g)
    return {"state": state, "param_groups": param_groups}


def load_checkpoint(model, path, optimizer=None, scheduler=None, map_location="cpu", flat=None, trusted=False):
    """Load a reference-format checkpoint.  Returns (epoch, missing, unexpected) where the two lists may only contain
    Hugging Face bookkeeping buffers (anything else raises, like strict=True); the reference resumes at `epoch + 1`
    (main_utils.py:131-134).  Only 'model' is layout-independent: a per-parameter 'optimizer' state (what the reference
    writes) is converted into the flat optimizer when `flat` (the FlatParams the optimizer was built on) is given; empty
    optimizer / scheduler entries are skipped.  trusted=False refuses pickles beyond tensors and the config namespace."""
    ckpt = _read(path, map_location, trusted)
    sd = _strip(ckpt["model"])
    res = model.load_state_dict(sd, strict=False)
    bad_missing = [k for k in res.missing_keys if k not in _HF_BOOKKEEPING]
    bad_unexpected = [k for k in res.unexpected_keys if k not in _HF_BOOKKEEPING]
    if bad_missing or bad_unexpected:
        raise RuntimeError(f"checkpoint does not match the model: missing {bad_missing[:8]}, unexpected {bad_unexpected[:8]}")
    ost = ckpt.get("optimizer") or None
    if optimizer is not None and ost and ost.get("param_groups"):
        n_saved = sum(len(g["params"]) for g in ost["param_groups"])
        n_here = sum(len(g["params"]) for g in optimizer.param_groups)
        if flat is not None and n_saved != n_here:
            reference_optimizer_state_to_flat(ost, model, flat, optimizer)
        else:
            optimizer.load_state_dict(ost)
    if scheduler is not None and ckpt.get("scheduler"):
        scheduler.load_state_dict(ckpt["scheduler"])
    if "eda_dropout_counter" in ckpt:            # (absent in checkpoints written by the reference)
        p0 = next(model.parameters(), None)
        if p0 is not None and p0.is_cuda:
            from . import attention
            attention.set_dropout_counter(p0.device, ckpt["eda_dropout_counter"])
            te = getattr(model, "text_encoder", None)
            if te is not None and ckpt.get("eda_text_dropout_counter") is not None:
                from . import roberta_fast
                roberta_fast.set_dropout_counter(te, ckpt["eda_text_dropout_counter"])
    return int(ckpt.get("epoch", 0)), list(res.missing_keys), list(res.unexpected_keys)


def save_checkpoint(model, path, optimizer=None, scheduler=None, epoch=0, config=None, flat=None):
    """Write ``path`` in the reference's layout (main_utils.py:149-166).  With `flat` the optimizer state is written per
    parameter in the reference's group order, so that the reference's `optimizer.load_state_dict` accepts it."""
    if optimizer is None:
        ost = {}
    elif flat is not None:
        ost = flat_optimizer_state_to_reference(model, flat, optimizer)
    else:
        ost = optimizer.state_dict()
    state = {"config": config, "save_path": path,
             "model": {"module." + k: v.detach().cpu() for k, v in model.state_dict().items()},
             "optimizer": ost,
             "scheduler": scheduler.state_dict() if scheduler is not None else {},
             "epoch": int(epoch)}
    p0 = next(model.parameters(), None)
    if p0 is not None and p0.is_cuda:
        # one extra key (ignored by the reference's loader): the library's dropout counter, so that a resumed run
        # continues the mask stream instead of replaying it
        from . import attention
        state["eda_dropout_counter"] = attention.get_dropout_counter(p0.device)
        te = getattr(model, "text_encoder", None)
        if te is not None:                       # (the frozen encoder's own counter: it may run on a second stream)
            from . import roberta_fast
            state["eda_text_dropout_counter"] = roberta_fast.get_dropout_counter(te)
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    torch.save(state, path)
    return state
