"""scamlgp_amd — MI355X-native batched GP inference behind the ScaML-GP model API.

Host side is Python on PyTorch-ROCm (tensor plumbing, streams, autograd glue); the hot
path is libscaml_hip.so (hand-written HIP for gfx950), bound through ctypes in ``_lib``.
"""
from ._lib import KIND_MATERN52, KIND_RBF, LIB_PATH  # noqa: F401
from . import ops  # noqa: F401

__all__ = ["KIND_RBF", "KIND_MATERN52", "LIB_PATH", "ops"]
