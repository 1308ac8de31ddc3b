"""Hot-path pieces of the reference's palette/utils.py: the HSV operators used by RegionEdit inside
the inference loop (palette/utils.py:257-295) and `normalize` (:126-127)."""
import ctypes

import torch
from torch.autograd import Function
from torch.amp import custom_fwd

from ._torch_glue import call, ptr, require, to_cuda

_u32 = ctypes.c_uint32


def normalize(tensor):
    return tensor / (tensor.norm(dim=-1, keepdim=True) + 1e-9)


class _rgb_to_hsv(Function):
    """palette/utils.py:258-274 (forward only)"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, input):
        input = to_cuda(input)
        prefix = input.shape[:-1]
        input = input.contiguous().view(-1, 3)
        n = input.shape[0]
        output = torch.empty(n, 3, device=input.device, dtype=input.dtype)
        call("pnr_rgb_to_hsv", _u32(n), ptr(require(input, torch.float32, "input")), ptr(output))
        return output.reshape(*prefix, 3)


rgb_to_hsv = _rgb_to_hsv.apply


class _hsv_to_rgb(Function):
    """palette/utils.py:278-293 (forward only)"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, input):
        input = to_cuda(input)
        prefix = input.shape[:-1]
        input = input.contiguous().view(-1, 3)
        n = input.shape[0]
        output = torch.empty(n, 3, device=input.device, dtype=input.dtype)
        call("pnr_hsv_to_rgb", _u32(n), ptr(require(input, torch.float32, "input")), ptr(output))
        return output.reshape(*prefix, 3)


hsv_to_rgb = _hsv_to_rgb.apply


def compute_RGB_histogram(colors_rgb, weights, bits_per_channel):
    """palette/utils.py:129-146: (colors_rgb [n,3] f32, weights [n] f32, bits) -> (bin_weights f64 [2^(3b)], bin_centers f32 [2^(3b),3]).
    NumPy in / NumPy out like the reference (whose implementation is CPU C++); torch tensors on the GPU are accepted and returned as is."""
    import numpy as np
    as_numpy = isinstance(colors_rgb, np.ndarray)
    c = torch.as_tensor(colors_rgb, dtype=torch.float32)
    w = torch.as_tensor(weights, dtype=torch.float32)
    assert c.ndim == 2 and c.shape[1] == 3 and w.ndim == 1 and len(c) == len(w) and 1 <= bits_per_channel <= 8
    c, w = to_cuda(c).contiguous(), to_cuda(w).contiguous()
    nb = 1 << (3 * bits_per_channel)
    bw = torch.empty(nb, dtype=torch.float64, device=c.device)
    bc = torch.empty(nb, 3, dtype=torch.float32, device=c.device)
    call("pnr_rgb_histogram", ptr(c), ptr(w), _u32(c.shape[0]), ctypes.c_int(bits_per_channel), ptr(bw), ptr(bc))
    return (bw.cpu().numpy(), bc.cpu().numpy()) if as_numpy else (bw, bc)
