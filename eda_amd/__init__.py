"""eda_amd -- MI355X (gfx950) native hot path of yanmin-wu/EDA.

Hand-written HIP kernels behind a C ABI (include/eda_hip.h, built into
eda_amd/csrc/libeda_hip.so) and the Python host layer that mirrors the
reference's operator / module interface for this path:

    eda_amd.ext               <-> pointnet2._ext          (9 native ops)
    eda_amd.pointnet2_utils   <-> pointnet2/pointnet2_utils.py
    eda_amd.pointnet2_modules <-> pointnet2/pointnet2_modules.py
    eda_amd.backbone_module   <-> models/backbone_module.py
    eda_amd.encoder_decoder_layers <-> models/encoder_decoder_layers.py
    eda_amd.bdetr             <-> models/bdetr.py
"""
__version__ = "0.1.0"

# ROCm 7.2 hazard: replays of a captured HIP graph through the runtime's pre-recorded
# AQL packets stop being equivalent to node-by-node launches after a host synchronisation between
# replays; training steps replayed that way drift.  Switch the optimisation off unless the user set
# the variable (it is read when the HIP runtime initialises, i.e. before the first device call).
import os as _os
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
