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


def _reload_knobs():
    """The library reads its EDA_* knobs once into one table (csrc/capi.hip); re-read them if it is loaded."""
    from eda_amd import _lib
    if _lib._lib is not None:
        _lib._lib.eda_reload_env()


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch, with setenv / delenv of an EDA_* knob followed by eda_reload_env() (and once more after
    the environment is restored), so a test can flip a library knob inside the process."""
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def setenv_reload(name, value, *a, **k):
        setenv(name, value, *a, **k)
        if name.startswith("EDA_"):
            _reload_knobs()

    def delenv_reload(name, *a, **k):
        delenv(name, *a, **k)
        if name.startswith("EDA_"):
            _reload_knobs()

    monkeypatch.setenv, monkeypatch.delenv = setenv_reload, delenv_reload
    yield monkeypatch
    monkeypatch.undo()
    _reload_knobs()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_ext
    oracle_ext.build()
    return oracle_ext
