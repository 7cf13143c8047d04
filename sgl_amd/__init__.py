"""sgl_amd -- MI355X-native SGAP pre-propagation (GraphOp.propagate + MessageOp aggregators) behind SGL's plugin API.

    from sgl_amd.operators.graph_op import LaplacianGraphOp, PprGraphOp
    from sgl_amd.operators.message_op import MeanMessageOp, LearnableWeightedMessageOp, ...
    from sgl_amd.models.homo import SGC, GAMLP, NAFS

Python (PyTorch-ROCm for device memory/streams) -> ctypes -> libsgl_hip.so (C ABI, include/sgl_hip.h) -> hand-written
HIP kernels for gfx950.  There is no CPU fallback: without the built library or without a GPU the hot path raises."""
from . import config  # noqa: F401

__version__ = "0.1.0"
