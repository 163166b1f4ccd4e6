"""MI355X-native SVD denoising hot path for the World-In-World WM server.

Host side (Python) of the drop-in described in DESIGN.md / INTEGRATION.md.  The compute runs in
`csrc/` (hand-written gfx950 HIP kernels behind the C ABI of `include/wiw_svd.h`); PyTorch is used
only for device memory, streams and torch.distributed plumbing.
"""
from .config import UNetConfig, SchedulerConfig  # noqa: F401

__all__ = ["UNetConfig", "SchedulerConfig"]
