"""Run-to-run reproducible training: the counterpart of the reference's `cudnn.deterministic = True`
(train_dist_mod.py:342-344).

`enable(True)` (or EDA_DETERMINISTIC=1 in the environment) switches every fp32-atomic gradient scatter of the library
to ordered per-owner sums (csrc/scatter_det.hip: group_points_grad, gather_points_grad, three_interpolate_grad, the
fused set-abstraction backward's d(features), the class-embedding weight gradient).  Everything else in the step already
is order-free: split contractions and attention partials are added in split order, weight gradients and LayerNorm
reductions go through slabs reduced in a fixed order, dropout is a counter hash.  What remains are the fp64 column
sums of the BatchNorm statistics (atomics over ~2000 workgroup partials): their order changes a sum by ~1e-16 relative,
which reaches an fp32 result only when the sum lies that close to a rounding boundary (~1e-3 per training step by
count) -- see DESIGN.md.  Cost of the mode: profiles/r05_bench_deterministic.json.
"""
from . import _lib


def enable(on=True):
    _lib.check(_lib.lib().eda_set_deterministic(1 if on else 0), "eda_set_deterministic")


def enabled():
    return bool(_lib.lib().eda_get_deterministic())
