"""Small helpers shared by the operator modules: pointer extraction, stream lookup, input checks."""
import ctypes

import torch

from . import _lib


# Per-call host cost matters on the drop-in operator path (a frame through the reference's run_cuda is ~110 of these calls; VERDICT round 4 measured 17.6 us
# for near_far_from_aabb through this wrapper against 4.2 us for the reference's pybind call): every entry point has its argtypes set once (_lib.load), so
# pointers, sizes and the stream travel as plain Python ints / floats -- no ctypes object per argument -- and the current stream's handle comes from
# torch's raw getter (no Stream object per call).
try:
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:      # a torch without the private getter: the public way
    _raw_stream = None


def stream_ptr():
    """The current HIP stream of the current device, as an int (0 = the null stream)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device address of a tensor as an int (None for None): entry points declare c_void_p, ctypes converts."""
    return None if t is None else t.data_ptr()


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
