"""The fields' bias-free MLP stacks for training as one HIP launch forward and one backward (csrc/mlp.hip).

`run_mlp(net, h, act)` is what `network._run` calls: the reference's layer loop (nerf/network.py:101-106, 113-118; palette/network.py:164-168,
240-262) `for l: h = net[l](h); if l != last: h = act(h)`.  On CUDA fp32 batches of at least MIN_ROWS rows with gradients enabled, 2 or 3
bias-free layers of width <= 64 and ReLU / ELU between them, it runs `pnr_mlp_forward` / `pnr_mlp_backward` (hidden activations recomputed
in the backward, weight gradients reduced deterministically); otherwise the plain torch loop.  Under fp16 autocast the fused path still
computes in fp32 (inputs cast up, fp32 out): faster than the half GEMM chain here and at least as accurate.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib
from ._torch_glue import call, ptr

MIN_ROWS = 8192
_ACT = {F.relu: 0, F.elu: 1}
enabled = True      # module switch (tests compare against the torch loop)


def _desc(dims, act):
    d = _lib.MlpDesc()
    d.n_layers = len(dims) - 1
    for i, v in enumerate(dims):
        d.dims[i] = v
    d.activation = act
    return d


class _FusedMLP(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)   # under autocast: fp32 in, fp32 out (at least the accuracy of the half GEMMs)
    def forward(ctx, x, act, *weights):
        dims = [weights[0].shape[1]] + [w.shape[0] for w in weights]
        desc = _desc(dims, act)
        lib = _lib.load()
        dev = x.device
        packed = torch.empty(int(lib.pnr_mlp_packed_bytes(ctypes.byref(desc))) // 4, dtype=torch.float32, device=dev)
        ws = [w.detach().contiguous() for w in weights]
        call("pnr_mlp_pack", ctypes.byref(desc), ptr(ws[0]), ptr(ws[1]), ptr(ws[2]) if len(ws) == 3 else None, ptr(packed))
        x2 = x.detach().reshape(-1, dims[0]).contiguous()
        B = x2.shape[0]
        y = torch.empty(B, dims[-1], dtype=torch.float32, device=dev)
        call("pnr_mlp_forward", ctypes.byref(desc), ptr(packed), ptr(x2), ctypes.c_uint32(B), ptr(y))
        ctx.save_for_backward(x2, packed)
        ctx.dims, ctx.act, ctx.x_shape = dims, act, x.shape
        return y.reshape(*x.shape[:-1], dims[-1])

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x2, packed = ctx.saved_tensors
        dims, act = ctx.dims, ctx.act
        desc = _desc(dims, act)
        lib = _lib.load()
        dev = x2.device
        B = x2.shape[0]
        dy2 = dy.reshape(-1, dims[-1]).contiguous().float()
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        n = len(dims) - 1
        dws = [torch.empty(dims[l + 1], dims[l], dtype=torch.float32, device=dev) if ctx.needs_input_grad[2 + l] else None for l in range(n)]
        nbytes = int(lib.pnr_mlp_backward_workspace_bytes(ctypes.byref(desc), B))
        ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
        call("pnr_mlp_backward", ctypes.byref(desc), ptr(packed), ptr(x2), ptr(dy2), ctypes.c_uint32(B), ptr(dx), ptr(dws[0]), ptr(dws[1]),
             ptr(dws[2]) if n == 3 else None, ptr(ws), ctypes.c_uint64(nbytes))
        return (dx.reshape(ctx.x_shape) if dx is not None else None, None, *dws)


def fusable(net, h, act):
    if not (enabled and h.is_cuda and h.dtype in (torch.float32, torch.float16) and torch.is_grad_enabled()):
        return False
    if h.dtype == torch.float16 and not torch.is_autocast_enabled():
        return False
    if act not in _ACT or len(net) not in (2, 3) or h.numel() // h.shape[-1] < MIN_ROWS:
        return False
    dims = [net[0].in_features] + [l.out_features for l in net]
    if max(dims) > 64 or any(l.bias is not None or l.weight.dtype != torch.float32 for l in net):
        return False
    tiles = [(d + 31) // 32 if i in (0, len(dims) - 1) else 2 for i, d in enumerate(dims)]
    packed = 2 * sum(tiles[i] * tiles[i + 1] for i in range(len(dims) - 1)) * 1024
    if (packed + 8 * 32 * 65) * 4 > 160 * 1024:      # weights (both orientations) + the backward's staging tiles must fit one CU's LDS
        return False
    return h.requires_grad or any(l.weight.requires_grad for l in net)


def run_mlp(net, h, act=F.relu):
    if fusable(net, h, act):
        return _FusedMLP.apply(h, _ACT[act], *[l.weight for l in net])
    for i, layer in enumerate(net):
        h = layer(h)
        if i != len(net) - 1:
            h = act(h, inplace=True)
    return h
