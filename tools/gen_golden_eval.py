"""tests/golden/eval_counts.npz: the counters of the REFERENCE's src/grounding_evaluator.py (imported from
/root/reference in this build container only; it needs torch + its own models/losses.py and utils/misc.py) on the seeded
end_points of tests/loss_fixtures.py + tests/eval_fixtures.py.  Only arrays are stored.
    python tools/gen_golden_eval.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import eval_fixtures as EF  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    for pkg in ("models", "utils"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    sys.modules["models"].losses = _load("models.losses", "/root/reference/models/losses.py")
    sys.modules["utils"].misc = _load("utils.misc", "/root/reference/utils/misc.py")
    return _load("ref_grounding_evaluator", "/root/reference/src/grounding_evaluator.py")


def main():
    ref = load_reference()
    out = {}
    for case, (seed, only_root, filt) in EF.CASES.items():
        ep = EF.make_end_points(seed)
        ev = ref.GroundingEvaluator(only_root=only_root, thresholds=[0.25, 0.5], topks=[1, 5, 10], prefixes=EF.PREFIXES,
                                    filter_non_gt_boxes=filt)
        for _ in range(2):                      # two batches accumulate
            for p in EF.PREFIXES:
                ev.evaluate(ep, p)
        keys = EF.counter_keys(ev)
        out[case + "_dets"] = np.array([float(ev.dets[k]) for k in keys], dtype=np.float64)
        out[case + "_gts"] = np.array([float(ev.gts[k]) for k in keys], dtype=np.float64)
        print(case, "Acc@0.25 top-1 last_ bbf:", ev.dets[("last_", 0.25, 1, "bbf")], "/", ev.gts[("last_", 0.25, 1, "bbf")],
              " top-10 bbs:", ev.dets[("last_", 0.25, 10, "bbs")])
    path = os.path.join(ROOT, "tests", "golden", "eval_counts.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
