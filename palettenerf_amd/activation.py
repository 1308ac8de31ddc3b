"""Density activation of the fields: exp() with a clamped gradient (what the reference calls trunc_exp, activation.py:5-17)."""
import torch
from torch.amp import custom_bwd, custom_fwd
from torch.autograd import Function


class _TruncExp(Function):
    """y = exp(x) in fp32; dy/dx is evaluated at clamp(x, -15, 15) so that a runaway logit cannot produce an infinite gradient."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, logits):
        ctx.save_for_backward(logits)
        return logits.exp()

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        (logits,) = ctx.saved_tensors
        return grad_out * logits.clamp(min=-15.0, max=15.0).exp()


def trunc_exp(x):
    return _TruncExp.apply(x)
