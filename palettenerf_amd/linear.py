"""Bias-free dense layer of the fields with an MI355X weight-gradient kernel.

Same module surface and state_dict entry (`weight`) as the `nn.Linear(in, out, bias=False)` layers the reference builds
(nerf/network.py:60-93, palette/network.py:60-153).  Forward and the input gradient stay library GEMMs; the weight gradient
dW = dY^T X -- a (<= 64) x (<= 64) result reduced over hundreds of thousands of samples, which the BLAS library runs at
1.0-1.5 ms per layer -- goes through `pnr_linear_wgrad` (csrc/linear.hip: the sample index is the MFMA k dimension).
"""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._torch_glue import call, ptr

MIN_ROWS = 2048      # below this the library GEMM is as fast and the launch count matters more
_DTYPE = {torch.float32: 0, torch.float16: 1}


def weight_grad(x, dy, out=None, accumulate=False):
    """x [B, in], dy [B, out] (fp32 or fp16, CUDA, contiguous) -> dW [out, in] fp32."""
    B, n_in = x.shape
    n_out = dy.shape[1]
    if x.dtype not in _DTYPE or dy.dtype not in _DTYPE:
        raise RuntimeError(f"weight_grad: unsupported dtypes {x.dtype}, {dy.dtype}")
    x, dy = x.contiguous(), dy.contiguous()
    if out is None:
        out = torch.empty(n_out, n_in, dtype=torch.float32, device=x.device)
        accumulate = False
    nbytes = int(_lib.load().pnr_linear_wgrad_workspace_bytes(B, n_in, n_out))
    ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=x.device)
    call("pnr_linear_wgrad", ptr(x), ctypes.c_int(_DTYPE[x.dtype]), ptr(dy), ctypes.c_int(_DTYPE[dy.dtype]), ctypes.c_uint32(B), ctypes.c_uint32(n_in),
         ctypes.c_uint32(n_out), ptr(out), ctypes.c_int(int(accumulate)), ptr(ws), ctypes.c_uint64(nbytes))
    return out


class _LinearNoBias(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight)          # under autocast this is the fp16 GEMM, as for nn.Linear

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dw = None
        dy2 = dy.reshape(-1, dy.shape[-1])
        if ctx.needs_input_grad[0]:
            dx = (dy2 @ weight.to(dy2.dtype)).reshape(x.shape).to(x.dtype)
        if ctx.needs_input_grad[1]:
            x2 = x.reshape(-1, x.shape[-1])
            dw = weight_grad(x2, dy2).to(weight.dtype)
        return dx, dw


def bias_grad(dy):
    """dy [B, out] (fp32 or fp16, CUDA) -> db [out] fp32 = column sums (pnr_linear_bgrad)."""
    B, n_out = dy.shape
    dy = dy.contiguous()
    out = torch.empty(n_out, dtype=torch.float32, device=dy.device)
    nbytes = int(_lib.load().pnr_linear_wgrad_workspace_bytes(B, 1, n_out))
    ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=dy.device)
    call("pnr_linear_bgrad", ptr(dy), ctypes.c_int(_DTYPE[dy.dtype]), ctypes.c_uint32(B), ctypes.c_uint32(n_out), ptr(out), ctypes.c_int(0), ptr(ws),
         ctypes.c_uint64(nbytes))
    return out


class _LinearBias(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight, bias)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dw = db = None
        dy2 = dy.reshape(-1, dy.shape[-1])
        if ctx.needs_input_grad[0]:
            dx = (dy2 @ weight.to(dy2.dtype)).reshape(x.shape).to(x.dtype)
        if ctx.needs_input_grad[1]:
            dw = weight_grad(x.reshape(-1, x.shape[-1]), dy2).to(weight.dtype)
        if ctx.needs_input_grad[2]:
            db = bias_grad(dy2).to(weight.dtype)
        return dx, dw, db


class Linear(nn.Linear):
    """nn.Linear whose weight (and bias) gradient runs on the HIP kernel when it pays (CUDA, >= MIN_ROWS rows, dims <= 64)."""

    def forward(self, x):
        if (x.is_cuda and torch.is_grad_enabled() and self.weight.requires_grad and self.in_features <= 64
                and self.out_features <= 64 and x.dtype in _DTYPE and x.numel() // x.shape[-1] >= MIN_ROWS):
            if self.bias is None:
                return _LinearNoBias.apply(x, self.weight)
            return _LinearBias.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)
