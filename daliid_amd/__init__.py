"""daliid_amd -- MI355X-native (gfx950) kernels under DaliID's Person-ReID hot path.

The package mirrors the reference's module surface for this path (Encoders / losses /
train_encodersKIT / getFeatures / validateModels / make_models / vit_pytorch) on top of a C-ABI
shared library of hand-written HIP kernels (``libdaliid_hip.so``, see ``include/daliid.h``).
There is no CPU or PyTorch fallback: importing an op without the built library, or calling it without
a gfx950 GPU, raises.
"""
__version__ = "0.1.0"
