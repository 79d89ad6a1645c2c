"""Kernels that stage through LDS-DMA must stay exact when ANOTHER queue's workgroups share their CUs.

Round 5 (profiles/r05_lds_dma_races.md): the DMA-staged product let a ds_read cross the barrier that re-arms its LDS stage;
exact on an idle GPU, wrong in one replay of three underneath the pipelined step's main graphs -- and in every replay once
the DMA-staged kernel is forced for all eligible shapes, which is what the second case pins.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"EDA_GEMM_SPLITK": "0", "EDA_GEMM_DMA": "1"}], ids=["default", "dma-staged-everywhere"])
def test_text_encoder_graph_is_exact_underneath_the_main_graphs(env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dbg_pipeline_gemm.py"), "6"], env=e, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "worst 0.0" in r.stdout
