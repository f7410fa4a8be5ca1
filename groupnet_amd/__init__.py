"""groupnet_amd — MI355X-native (gfx950) implementation of GroupNet's multiscale hypergraph
message-passing hot path, behind the reference's own nn.Module API.

    from groupnet_amd.MS_HGNN_batch import MS_HGNN_oridinary, MS_HGNN_hyper, MLP

All compute is in libgroupnet_hip.so (hand-written HIP kernels, C ABI in include/groupnet_hip.h);
this package is the ctypes binding plus the drop-in modules.  No CPU fallback exists.
"""
from . import _lib, ops  # noqa: F401
from .MS_HGNN_batch import (MLP, MLP_dict_softmax, MS_HGNN_hyper, MS_HGNN_oridinary, edge_aggregation,
                            set_noise_mode)
from .past_encoder import FutureEncoder, PastEncoder, PositionalAgentEncoding

__all__ = ["MLP", "MLP_dict_softmax", "MS_HGNN_hyper", "MS_HGNN_oridinary", "edge_aggregation", "ops",
           "set_noise_mode", "PastEncoder", "FutureEncoder", "PositionalAgentEncoding"]
__version__ = "0.1.0"
