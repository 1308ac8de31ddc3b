"""Small helpers shared by the operator modules: pointer extraction, stream lookup, input checks."""
import ctypes

import torch

from . import _lib


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("palettenerf_amd: expected a CUDA(HIP) tensor, got a CPU tensor (no CPU fallback exists)")


def require(t, dtype, name):
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be a {dtype} tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    return t


_profile = None  # name -> list of (start_event, end_event, units); see profile_kernels()


def profile_kernels(names=None):
    """Enable (list of C-ABI entry names) or disable (None) HIP-event timing of individual launches.
    Events are recorded on torch's current stream, the stream the kernels are launched on."""
    global _profile
    _profile = None if names is None else {n: [] for n in names}
    return _profile


def call(name, *args, units=0):
    if _profile is not None and name in _profile:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call(name, *args, stream_ptr())
        e1.record()
        _profile[name].append((e0, e1, units))
        return
    _lib.call(name, *args, stream_ptr())


def to_cuda(t):
    return t if t.is_cuda else t.cuda()
