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
import os

import torch

_HF_BOOKKEEPING = ("text_encoder.embeddings.position_ids", "text_encoder.embeddings.token_type_ids")


def _strip(sd):
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def load_checkpoint(model, path, optimizer=None, scheduler=None, map_location="cpu"):
    """Load a reference-format checkpoint.  Returns (epoch, missing, unexpected) where the two lists may
    only contain Hugging Face bookkeeping buffers (anything else raises, like strict=True)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    sd = _strip(ckpt["model"])
    res = model.load_state_dict(sd, strict=False)
    bad_missing = [k for k in res.missing_keys if k not in _HF_BOOKKEEPING]
    bad_unexpected = [k for k in res.unexpected_keys if k not in _HF_BOOKKEEPING]
    if bad_missing or bad_unexpected:
        raise RuntimeError(f"checkpoint does not match the model: missing {bad_missing[:8]}, unexpected {bad_unexpected[:8]}")
    if optimizer is not None and "optimizer" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])
    if scheduler is not None and "scheduler" in ckpt:
        scheduler.load_state_dict(ckpt["scheduler"])
    if "eda_dropout_counter" in ckpt:            # (absent in checkpoints written by the reference)
        p0 = next(model.parameters(), None)
        if p0 is not None and p0.is_cuda:
            from . import attention
            attention.set_dropout_counter(p0.device, ckpt["eda_dropout_counter"])
    return int(ckpt.get("epoch", 0)), list(res.missing_keys), list(res.unexpected_keys)


def save_checkpoint(model, path, optimizer=None, scheduler=None, epoch=0, config=None):
    """Write ``path`` in the reference's layout (main_utils.py:149-166)."""
    state = {"config": config, "save_path": path,
             "model": {"module." + k: v.detach().cpu() for k, v in model.state_dict().items()},
             "optimizer": optimizer.state_dict() if optimizer is not None else {},
             "scheduler": scheduler.state_dict() if scheduler is not None else {},
             "epoch": int(epoch)}
    p0 = next(model.parameters(), None)
    if p0 is not None and p0.is_cuda:
        # one extra key (ignored by the reference's loader): the library's dropout counter, so that a resumed run
        # continues the mask stream instead of replaying it
        from . import attention
        state["eda_dropout_counter"] = attention.get_dropout_counter(p0.device)
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    torch.save(state, path)
    return state
