"""Evaluation post-processing parity: every counter of eda_amd.grounding_evaluator.GroundingEvaluator
equals the reference's src/grounding_evaluator.py run on the same seeded end_points in the build container
(tests/golden/eval_counts.npz, tools/gen_golden_eval.py): position and semantic alignment, root only / every annotated
object, the detected-box filter, accumulation over batches, the `last_` break-downs."""
import os

import numpy as np
import pytest
import torch

import eval_fixtures as EF

GOLD = os.path.join(os.path.dirname(__file__), "golden", "eval_counts.npz")


def _run(dev):
    from eda_amd.grounding_evaluator import GroundingEvaluator
    g = np.load(GOLD)
    for case, (seed, only_root, filt) in EF.CASES.items():
        ep = EF.make_end_points(seed)
        ep = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in ep.items()}
        ev = GroundingEvaluator(only_root=only_root, thresholds=[0.25, 0.5], topks=[1, 5, 10], prefixes=EF.PREFIXES,
                                filter_non_gt_boxes=filt)
        for _ in range(2):
            for p in EF.PREFIXES:
                ev.evaluate(ep, p)
        keys = EF.counter_keys(ev)
        assert len(keys) == len(g[case + "_dets"])
        dets = np.array([float(ev.dets[k]) for k in keys])
        gts = np.array([float(ev.gts[k]) for k in keys])
        bad = [(k, d, e) for k, d, e in zip(keys, dets, g[case + "_dets"]) if d != e]
        assert not bad, (case, bad[:5])
        np.testing.assert_allclose(gts, g[case + "_gts"], rtol=0, atol=1e-12, err_msg=case)
        assert dets.sum() > 0 and (dets < gts - 0.5).any()          # the fixture has hits AND misses


def test_evaluator_counters_equal_the_reference_cpu():
    _run("cpu")


@pytest.mark.gpu
def test_evaluator_counters_equal_the_reference_gpu():
    _run("cuda")


def test_print_stats_and_reset():
    from eda_amd.grounding_evaluator import GroundingEvaluator
    ev = GroundingEvaluator(prefixes=["last_"])
    ep = EF.make_end_points(21)
    ev.evaluate(ep, "last_")
    ev.print_stats()
    assert ev.gts[("last_", 0.25, 1, "bbf")] == 8
    ev.reset()
    assert ev.dets[("last_", 0.25, 1, "bbf")] == 0 and ev.gts["vd"] == 1e-14
    ev.synchronize_between_processes()          # no process group: a no-op
