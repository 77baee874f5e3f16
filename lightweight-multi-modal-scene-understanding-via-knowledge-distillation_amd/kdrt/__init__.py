"""kdrt -- host runtime of the MI355X-native KD training path.

Python here is plumbing only (device memory via PyTorch's allocator, HIP streams, autograd
bookkeeping, torch.distributed/RCCL).  Every arithmetic step of the hot path runs in
``csrc/libkd_hip.so`` (hand-written gfx950 kernels behind the C ABI in ``include/kd_hip.h``).
There is no CPU or stock-PyTorch fallback: if the library is missing, importing ``kdrt.lib``
raises, and every op raises on non-GPU tensors.
"""
from .lib import lib, KDError, require_gpu_tensor  # noqa: F401
