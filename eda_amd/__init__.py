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
