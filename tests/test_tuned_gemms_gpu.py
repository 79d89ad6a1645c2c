"""bench.py runs the library GEMMs with TunableOp-selected solutions (eda_amd/gemm_tuning.py,
results shipped in eda_amd/tuned/).  The parity tests of the model against the reference goldens
must hold with those selections too: re-run them in a child process that enables the shipped
results (EDA_TUNED_GEMMS=1, see conftest.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shipped_results_load_on_this_stack():
    import torch
    from eda_amd import gemm_tuning
    assert os.path.exists(gemm_tuning.SHIPPED)
    code = ("import sys; sys.path.insert(0, %r); import torch; from eda_amd import gemm_tuning; "
            "f, ok = gemm_tuning.enable(online=False); "
            "a = torch.randn(2048, 288, device='cuda'); w = torch.randn(288, 288, device='cuda'); "
            "torch.testing.assert_close(a @ w.t(), (a.double() @ w.double().t()).float(), rtol=1e-4, atol=1e-3); "
            "print('LOADED' if ok else 'REJECTED', len(torch.cuda.tunable.get_results()))" % os.path.dirname(HERE))
    out = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True, timeout=300).stdout
    # a results file recorded with other library versions is rejected by TunableOp's validators and
    # the default selection is used: allowed, but on the image this repo targets it must load
    assert "LOADED" in out or "REJECTED" in out, out
    if torch.version.hip and torch.__version__.startswith("2.10"):
        assert "LOADED" in out, out


def test_model_parity_holds_with_tuned_gemms():
    env = dict(os.environ, EDA_TUNED_GEMMS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu",
                        os.path.join(HERE, "test_model_gpu.py"), os.path.join(HERE, "test_attention.py"),
                        os.path.join(HERE, "test_wgrad_queue_gpu.py")],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
