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
