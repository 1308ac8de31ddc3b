"""Real spherical-harmonics direction encoder with the reference's operator surface (`_sh_encoder`, `sh_encode`, `SHEncoder`;
shencoder/sphere_harmonics.py), evaluated by pnr_sh_encode_{forward,backward} (degree 1..8, fp32)."""
import ctypes

import torch
import torch.nn as nn
from torch.amp import custom_bwd, custom_fwd
from torch.autograd import Function

from ._torch_glue import call, ptr, require

_u32 = ctypes.c_uint32


class _sh_encoder(Function):
    """inputs [B,3] -> [B, degree^2]; with calc_grad_inputs the analytic Jacobian [B, 3*degree^2] is kept for backward."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)  # directions are always encoded in fp32
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        dirs = require(inputs.contiguous(), torch.float32, "inputs")
        n, dim = dirs.shape
        width = degree * degree
        basis = torch.empty(n, width, dtype=torch.float32, device=dirs.device)
        jac = torch.empty(n, dim * width, dtype=torch.float32, device=dirs.device) if calc_grad_inputs else None
        call("pnr_sh_encode_forward", ptr(dirs), ptr(basis), _u32(n), _u32(dim), _u32(degree), ptr(jac))
        ctx.save_for_backward(dirs, jac)
        ctx.shape = (n, dim, degree)
        return basis

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        dirs, jac = ctx.saved_tensors
        if jac is None:
            return None, None, None
        n, dim, degree = ctx.shape
        grad_dirs = torch.zeros_like(dirs)  # the kernel accumulates into it
        call("pnr_sh_encode_backward", ptr(require(grad.contiguous(), torch.float32, "grad")), ptr(dirs), _u32(n), _u32(dim), _u32(degree), ptr(jac),
             ptr(grad_dirs))
        return grad_dirs, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        assert input_dim == 3, "SH encoder only support input dim == 3"
        assert 0 < degree <= 8, "SH encoder only supports degree in [1, 8]"
        self.input_dim, self.degree, self.output_dim = input_dim, degree, degree ** 2

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        """inputs [..., 3] in [-size, size] -> [..., degree^2]"""
        scaled = inputs / size
        flat = scaled.reshape(-1, self.input_dim)
        return sh_encode(flat, self.degree, flat.requires_grad).reshape(list(scaled.shape[:-1]) + [self.output_dim])
