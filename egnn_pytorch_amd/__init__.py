"""egnn_pytorch_amd -- MI355X (gfx950) native drop-in for `egnn_pytorch.EGNN.forward`.

    from egnn_pytorch_amd import EGNN, EGNN_Network      # same constructor / forward / state_dict

The compute path is libegnn_hip.so (hand-written HIP kernels behind the C ABI of
include/egnn_hip.h).  PyTorch is used for device memory, streams and torch.distributed only.
"""
from .layer import EGNN, EGNN_Network, CoorsNorm, exact_arithmetic
from ._ops import phase_timer, check_range
from ._abi import EGNNHipError, EGNNRangeError
from . import sharding
from .graph import graphed

__all__ = ["EGNN", "EGNN_Network", "CoorsNorm", "phase_timer", "sharding", "graphed", "check_range", "EGNNHipError",
           "EGNNRangeError", "exact_arithmetic"]
__version__ = "0.1.0"
