"""nsgp -- MI355X-native engine behind the models.* surface of Stansfash/nonstationary-precip.

Python host on PyTorch-ROCm (memory, streams, autograd edges, torch.distributed) driving hand-written
gfx950 HIP kernels through the C ABI of include/nsgp.h (ctypes).  No CPU fallback: importing is safe
anywhere, but every op needs the built library and CUDA tensors (`BackendError` otherwise).
"""
from ._lib import BackendError, LIB_PATH, declared_symbols, load as load_library  # noqa: F401
from . import ops  # noqa: F401

__all__ = ['BackendError', 'LIB_PATH', 'declared_symbols', 'load_library', 'ops']
