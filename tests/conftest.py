import os as _os
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # ROCm 7.2 graph-replay hazard, DESIGN.md §4
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("EDA_TUNED_GEMMS") == "1":
        # tests/test_tuned_gemms_gpu.py re-runs the model parity tests in a child process with the
        # TunableOp-selected library GEMMs that bench.py uses (eda_amd/gemm_tuning.py)
        import torch
        if torch.cuda.is_available():
            from eda_amd import gemm_tuning
            gemm_tuning.enable(online=False)


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _poisoned_allocator():
    """GPU sessions start with the caching allocator's free blocks full of NaNs (small and large
    pools), so a kernel or host routine that reads memory it never wrote shows up as a failure
    instead of passing on the zero pages a fresh process usually gets."""
    import torch
    if torch.cuda.is_available():
        junk = []
        for nbytes in (512, 4096, 65536, 1 << 20, 8 << 20, 64 << 20, 256 << 20):
            for _ in range(24 if nbytes <= (1 << 20) else 6):
                junk.append(torch.full((nbytes // 4,), float("nan"), device="cuda"))
        torch.cuda.synchronize()
        del junk
    yield


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_ext
    oracle_ext.build()
    return oracle_ext
