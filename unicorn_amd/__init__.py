"""unicorn_amd — MI355X-native (gfx950) implementation of Unicorn's per-frame inference hot path.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed plumbing); all
arithmetic of the path runs in hand-written HIP kernels behind the C-ABI library
``unicorn_amd/lib/libunicorn_hip.so`` (see include/unicorn_hip.h).  There is NO CPU / PyTorch
fallback: importing the ops without the built library, or calling them without a GPU, raises.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
