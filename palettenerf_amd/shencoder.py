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


class _sh_encode_cat(Function):
    """torch.cat([sh_encode(dirs, degree), tail], dim=-1) as one launch (pnr_sh_encode_cat_forward); no gradient for dirs, the tail's
    gradient is the column slice of the output's."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, dirs, tail, degree):
        dirs = require(dirs.contiguous(), torch.float32, "inputs")
        tail = require(tail.contiguous(), torch.float32, "tail")
        n, t = tail.shape
        out = torch.empty(n, degree * degree + t, dtype=torch.float32, device=dirs.device)
        call("pnr_sh_encode_cat_forward", ptr(dirs), ptr(tail), _u32(t), ptr(out), _u32(n), _u32(degree))
        ctx.width = degree * degree
        return out

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        return None, grad[:, ctx.width:], None


class _sigma_geo_cat(Function):
    """(trunc_exp(h[:, 0]), torch.cat([sh_encode(dirs, degree), h[:, 1:]], -1)) -- the seam between sigma_net and color_net (nerf/network.py:109-121) --
    as one launch forward and one backward: autograd's own backward of the select and the slice zero-fills two [B, hw] tensors, copies into both and adds
    them (~80 us on a 627 k-sample training batch)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, h, dirs, degree):
        h = require(h.contiguous(), torch.float32, "h")
        dirs = require(dirs.contiguous(), torch.float32, "inputs")
        n, hw = h.shape
        sigma = torch.empty(n, dtype=torch.float32, device=h.device)
        out = torch.empty(n, degree * degree + hw - 1, dtype=torch.float32, device=h.device)
        call("pnr_sigma_geo_cat_forward", ptr(h), _u32(hw), ptr(dirs), _u32(degree), _u32(n), ptr(sigma), ptr(out))
        ctx.save_for_backward(h)
        ctx.degree = degree
        return sigma, out

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dsigma, dout):
        (h,) = ctx.saved_tensors
        n, hw = h.shape
        grad_h = torch.empty_like(h)
        dsigma = None if dsigma is None else require(dsigma.contiguous(), torch.float32, "dsigma")
        dout = None if dout is None else require(dout.contiguous(), torch.float32, "dout")
        call("pnr_sigma_geo_cat_backward", ptr(h), _u32(hw), ptr(dsigma), ptr(dout), _u32(ctx.degree), _u32(n), ptr(grad_h))
        return grad_h, None, None


def sigma_geo_cat(encoder, h, dirs):
    """sigma, colour-net input of the NeRF field from sigma_net's output h [B, 1 + geo] and the view directions; the fused launch when it can be
    (see sh_encode_cat), the reference's composition otherwise."""
    from .activation import trunc_exp
    ok = (isinstance(encoder, SHEncoder) and h.is_cuda and h.ndim == 2 and dirs.ndim == 2 and dirs.shape[0] == h.shape[0] and h.shape[0] > 0 and h.shape[1] >= 2
          and not dirs.requires_grad and encoder.output_dim + h.shape[1] - 1 <= 64 and h.dtype == torch.float32 and dirs.dtype == torch.float32
          and not torch.is_autocast_enabled())
    if not ok:
        return trunc_exp(h[..., 0]), sh_encode_cat(encoder, dirs, h[..., 1:])
    return _sigma_geo_cat.apply(h, dirs, encoder.degree)


def sh_encode_cat(encoder, dirs, tail):
    """`torch.cat([encoder(dirs), tail], dim=-1)` for an SHEncoder: one launch when it can be (CUDA, [B,3] directions that need no gradient,
    degree^2 + tail columns <= 64), the plain composition otherwise."""
    ok = (isinstance(encoder, SHEncoder) and dirs.is_cuda and dirs.ndim == 2 and tail.ndim == 2 and dirs.shape[0] == tail.shape[0] and dirs.shape[0] > 0
          and not dirs.requires_grad and 0 < tail.shape[1] and encoder.output_dim + tail.shape[1] <= 64
          and dirs.dtype in (torch.float32, torch.float16) and tail.dtype in (torch.float32, torch.float16))
    if ok and not torch.is_autocast_enabled() and not (dirs.dtype == tail.dtype == torch.float32):   # under autocast both are cast up to fp32
        ok = False
    if not ok:
        return torch.cat([encoder(dirs), tail], dim=-1)
    return _sh_encode_cat.apply(dirs, tail, encoder.degree)


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
