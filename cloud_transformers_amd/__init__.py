"""MI355X-native Multi-Headed Cloud Transform (Splat / Slice / multihead_ct*).

Python + PyTorch-ROCm host code over hand-written HIP kernels (libcloudct.so,
C ABI in include/cloudct.h).  No CPU fallback: ops raise on non-HIP tensors.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
