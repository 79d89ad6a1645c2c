"""Deterministic weights and inputs shared by tools/gen_golden_model.py (which
runs the REFERENCE's Python layers in the build container) and the tests (which
run this repo's layers): both sides fill their modules from ``det_state`` and
feed ``make_*`` inputs, so a fixture only has to store the reference's outputs.
"""
import zlib

import numpy as np
import torch


def det_tensor(key, shape, dtype=torch.float32, seed=0):
    rng = np.random.default_rng(zlib.crc32(key.encode()) + 7919 * seed)
    shape = tuple(shape)
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    if key.endswith("running_var"):
        a = rng.uniform(0.5, 1.5, shape)
    elif key.endswith("running_mean"):
        a = rng.normal(0, 0.1, shape)
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        a = rng.normal(0, 1.0 / np.sqrt(fan_in), shape)
    elif key.endswith("weight"):           # 1-D weight: a norm layer's scale
        a = rng.uniform(0.5, 1.5, shape)
    else:                                  # biases
        a = rng.normal(0, 0.05, shape)
    # spread and centre the seed-objectness logits so the query top-k has clear gaps
    if key.endswith("points_obj_cls.conv3.weight"):
        a = a * 40.0
    if key.endswith("points_obj_cls.conv3.bias"):
        a = a * 0 - 13.0
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(dtype)


def fill_det_state(module, seed=0, skip_prefix=()):
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if any(k.startswith(p) for p in skip_prefix):
            new[k] = v
        elif k.endswith("position_ids") or v.dtype in (torch.int64, torch.int32) and not k.endswith("num_batches_tracked"):
            new[k] = v
        else:
            new[k] = det_tensor(k, v.shape, v.dtype, seed)
    module.load_state_dict(new)
    return module


def make_cloud(seed, b, n, with_color=True):
    from eda_amd import synthetic
    pc = synthetic.batch([seed * 100 + i for i in range(b)], n_points=n, with_color=with_color)
    return torch.from_numpy(pc)


def make_feats(seed, *shape, scale=1.0):
    rng = np.random.default_rng(555 + seed)
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def make_mask(seed, b, n, min_valid=1):
    """key_padding_mask (True = ignore): a valid prefix of random length."""
    rng = np.random.default_rng(777 + seed)
    lens = rng.integers(min_valid, n + 1, b)
    lens[0] = n
    return torch.from_numpy(np.arange(n)[None, :] >= lens[:, None])


def small_roberta(layers=1):
    from transformers import RobertaConfig, RobertaModel
    cfg = RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1,
                        pad_token_id=1, num_hidden_layers=layers)
    m = RobertaModel(cfg)
    m.eval()
    return m


def full_model_inputs(seed, b=2, n=4096, max_len=16):
    from eda_amd import synthetic
    ids, am = synthetic.utterance_tokens(seed, b, max_len=max_len)
    boxes, bmask, cls = synthetic.detected_boxes(seed, b)
    return {
        "point_clouds": make_cloud(seed, b, n),
        "tokenized": {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(am)},
        "det_boxes": torch.from_numpy(boxes),
        "det_bbox_label_mask": torch.from_numpy(bmask),
        "det_class_ids": torch.from_numpy(cls),
    }


# ---- compact fixtures: big float tensors are stored as a seeded subsample + sums ----
_SUB = 4096
_BIG = 16384


def _sub_index(name, numel):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return rng.integers(0, numel, _SUB)


def pack(arrays):
    """dict of tensors/arrays -> dict for np.savez (subsampling big float arrays)."""
    out = {}
    for k, v in arrays.items():
        a = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        if a.dtype.kind == "f" and a.size > _BIG:
            flat = a.reshape(-1)
            out[k + "__sub"] = flat[_sub_index(k, flat.size)]
            out[k + "__stats"] = np.array([flat.astype(np.float64).sum(),
                                           np.abs(flat.astype(np.float64)).sum(), flat.size])
            out[k + "__shape"] = np.array(a.shape)
        else:
            out[k] = a
    return out


def names(golden):
    return sorted({k.split("__")[0] for k in golden.files})


def _report(name, a, g, rtol, atol):
    """EDA_TEST_VERBOSE=1: print how much of the allowed error each tensor actually uses."""
    import os
    if os.environ.get("EDA_TEST_VERBOSE") and g.dtype.kind == "f" and g.size:
        err = np.abs(a.astype(np.float64) - g.astype(np.float64))
        allowed = atol + rtol * np.abs(g.astype(np.float64))
        print(f"  TOL {name}: used {float((err / np.maximum(allowed, 1e-300)).max()):.3f} of rtol={rtol} atol={atol}; "
              f"max err {float(err.max()):.3e}, max |g| {float(np.abs(g).max()):.3e}")


def assert_matches(golden, name, value, rtol=1e-4, atol=None, atol_rel=1e-5):
    """Compare `value` (tensor) with the packed fixture entry `name`: element-wise
    |value - golden| <= rtol * |golden| + atol, with atol = atol_rel * max|golden| unless given.
    Default = the north star's 1e-4 relative per element plus 1e-5 of the tensor's scale (fp32
    accumulation noise on elements that cancel to ~0; the goldens themselves are fp32 results of the
    reference on a CPU).  Measured use of this budget on MI355X (EDA_TEST_VERBOSE=1): <= 35 %."""
    a = value.detach().cpu().numpy() if torch.is_tensor(value) else np.asarray(value)
    if atol is None:
        key = name if name in golden.files else name + "__sub"
        gg = golden[key]
        atol = atol_rel * float(np.abs(gg).max()) if gg.dtype.kind == "f" and gg.size else 0.0
    if name in golden.files:
        g = golden[name]
        assert tuple(g.shape) == tuple(a.shape), (name, g.shape, a.shape)
        if g.dtype.kind in "iub":
            assert (g == a).all(), name
        else:
            _report(name, a, g, rtol, atol)
            np.testing.assert_allclose(a, g, rtol=rtol, atol=atol, err_msg=name)
        return
    assert tuple(golden[name + "__shape"]) == tuple(a.shape), (name, golden[name + "__shape"], a.shape)
    flat = a.reshape(-1)
    _report(name, flat[_sub_index(name, flat.size)], golden[name + "__sub"], rtol, atol)
    np.testing.assert_allclose(flat[_sub_index(name, flat.size)], golden[name + "__sub"],
                               rtol=rtol, atol=atol, err_msg=name)
    s, sa, n = golden[name + "__stats"]
    assert n == flat.size
    np.testing.assert_allclose(np.abs(flat.astype(np.float64)).sum(), sa, rtol=1e-4, err_msg=name + " abs-sum")
    np.testing.assert_allclose(flat.astype(np.float64).sum(), s, rtol=1e-3, atol=1e-3 * sa / max(n, 1) * 100,
                               err_msg=name + " sum")
