"""Library-GEMM selection for the path's fp32 GEMMs (PyTorch TunableOp).

The forward / input-gradient GEMMs of the pointwise layers stay library calls (hipBLASLt /
rocBLAS through torch).  For these small, oddly shaped fp32 problems (2048..8192 x 288 x 288,
1 048 576 x 64 x 64, ...) the libraries' default heuristics are often not the fastest solution
they contain: letting TunableOp time the candidates per shape is worth ~2 ms of a 32 ms step
on MI355X.  `enable()` switches TunableOp on, loads the results shipped in
``eda_amd/tuned/tunableop_gfx950.csv`` (recorded with `python bench.py --gemm-tuning record` on
gfx950, ROCm 7.2 / torch 2.10; TunableOp rejects the file when its library-version validators
differ) and, with ``online=True``, tunes any shape that is not in the file the first time it is
seen -- which must happen before a HIP-graph capture (bench.py's eager warm-up steps do that).
Newly tuned results go to `scratch` (TunableOp writes them itself), never into the package.
"""
import os
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SHIPPED = os.path.join(HERE, "tuned", "tunableop_gfx950.csv")


def enable(online=True, scratch=None, use_shipped=True, max_tuning_ms=30, max_iterations=100):
    """Returns (results file TunableOp writes to, whether the shipped results were accepted)."""
    tn = torch.cuda.tunable
    tn.enable(True)
    tn.tuning_enable(bool(online))
    tn.set_max_tuning_duration(int(max_tuning_ms))
    tn.set_max_tuning_iterations(int(max_iterations))
    if scratch is None:
        scratch = os.path.join(tempfile.gettempdir(), "eda_tunableop_%d_.csv" % os.getpid())
    tn.set_filename(scratch, insert_device_ordinal=True)
    loaded = False
    if use_shipped and os.path.exists(SHIPPED):
        try:
            loaded = bool(tn.read_file(SHIPPED))
        except Exception:                      # a results file from other library versions is simply not used
            loaded = False
    return tn.get_filename(), loaded


def disable():
    tn = torch.cuda.tunable
    tn.tuning_enable(False)
    tn.enable(False)
