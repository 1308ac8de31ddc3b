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


def _al(t):
    """`t`, contiguous and starting on a 16-byte boundary: the launches move their tiles as 16-byte requests (pnr_mlp_* refuse other arrays).  torch allocations are
    aligned; a contiguous view that starts inside one (x[1:] of a 3-wide tensor) is copied."""
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone()

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
    def forward(ctx, x, act, *weights):   # act: _ACT code, | _lib.MLP_OUT_SIGMOID for torch.sigmoid on the output
        dims = [weights[0].shape[1]] + [w.shape[0] for w in weights]
        desc = _desc(dims, act)
        lib = _lib.load()
        dev = x.device
        packed = torch.empty(int(lib.pnr_mlp_packed_bytes(ctypes.byref(desc))) // 4, dtype=torch.float32, device=dev)
        ws = [w.detach().contiguous() for w in weights]
        call("pnr_mlp_pack", ctypes.byref(desc), ptr(ws[0]), ptr(ws[1]), ptr(ws[2]) if len(ws) == 3 else None, ptr(packed))
        x2 = _al(x.detach().reshape(-1, dims[0]))
        B = x2.shape[0]
        y = torch.empty(B, dims[-1], dtype=torch.float32, device=dev)
        call("pnr_mlp_forward", ctypes.byref(desc), ptr(packed), ptr(x2), ctypes.c_uint32(B), ptr(y))
        ctx.save_for_backward(x2, packed, y if act & _lib.MLP_OUT_SIGMOID else None)
        ctx.dims, ctx.act, ctx.x_shape = dims, act, x.shape
        return y.reshape(*x.shape[:-1], dims[-1])

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x2, packed, y = ctx.saved_tensors
        dims, act = ctx.dims, ctx.act
        desc = _desc(dims, act)
        lib = _lib.load()
        dev = x2.device
        B = x2.shape[0]
        dy2 = _al(dy.reshape(-1, dims[-1]).float())
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        n = len(dims) - 1
        dws = [torch.empty(dims[l + 1], dims[l], dtype=torch.float32, device=dev) if ctx.needs_input_grad[2 + l] else None for l in range(n)]
        nbytes = int(lib.pnr_mlp_backward_workspace_bytes(ctypes.byref(desc), B))
        ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
        call("pnr_mlp_backward", ctypes.byref(desc), ptr(packed), ptr(x2), ptr(y), ptr(dy2), ctypes.c_uint32(B), ptr(dx), ptr(dws[0]), ptr(dws[1]),
             ptr(dws[2]) if n == 3 else None, ptr(ws), ctypes.c_uint64(nbytes))
        return (dx.reshape(ctx.x_shape) if dx is not None else None, None, *dws)


def fusable(net, h, act):
    if not (enabled and h.is_cuda and h.dtype in (torch.float32, torch.float16) and torch.is_grad_enabled()):
        return False
    if h.dtype == torch.float16 and not torch.is_autocast_enabled():
        return False
    if act not in _ACT or len(net) not in (2, 3) or not MIN_ROWS <= h.numel() // h.shape[-1] < 2 ** 26:
        return False
    dims = [net[0].in_features] + [l.out_features for l in net]
    if max(dims) > 64 or any(l.bias is not None or l.weight.dtype != torch.float32 for l in net):
        return False
    tiles = [(d + 31) // 32 if i in (0, len(dims) - 1) else 2 for i, d in enumerate(dims)]
    packed = 2 * sum(tiles[i] * tiles[i + 1] for i in range(len(dims) - 1)) * 1024
    if (packed + 8 * 32 * 65) * 4 > 160 * 1024:      # weights (both orientations) + the backward's staging tiles must fit one CU's LDS
        return False
    return h.requires_grad or any(l.weight.requires_grad for l in net)


def run_mlp(net, h, act=F.relu, out=None):
    """out: None, or torch.sigmoid applied to the stack's output (the colour heads: nerf/network.py:122, palette/network.py:245,254) -- inside
    the same two launches on the fused path."""
    if out not in (None, torch.sigmoid):
        raise ValueError("run_mlp: the output activation is None or torch.sigmoid")
    if fusable(net, h, act):
        return _FusedMLP.apply(h, _ACT[act] | (_lib.MLP_OUT_SIGMOID if out is not None else 0), *[l.weight for l in net])
    for i, layer in enumerate(net):
        h = layer(h)
        if i != len(net) - 1:
            h = act(h, inplace=True)
    return h if out is None else out(h)


class _EncodeMLP(torch.autograd.Function):
    """hash-grid lookup -> (optional tail columns) -> bias-free MLP with the encoder output kept level-major end to end
    (pnr_grid_encode_forward -> pnr_mlp_forward_lm; backward pnr_mlp_backward_lm -> pnr_grid_encode_backward_binned): no [L,B,C] <-> [B,L*C]
    copies, no torch.cat.  x01 [B,3] in [0,1] (no gradient), tail [B,t] or None (no gradient)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x01, embeddings, offsets, meta, tail, act, enc_pre, *weights):
        # enc_pre: the encoder's raw level-major output [L, B, 2] at x01 when the caller has looked it up already (pair lookup); an explicit,
        # non-differentiable input (the table's gradient flows through `embeddings` as always), saved for backward like a lookup made here
        from .gridencoder import _u32, _f32, _int     # ctypes aliases
        S, H, gridtype, align_corners = meta
        x01 = x01.contiguous()
        B = x01.shape[0]
        L = offsets.shape[0] - 1
        dev = x01.device
        emb = embeddings.detach().contiguous()
        if enc_pre is not None:
            enc = _al(enc_pre.detach())
            if enc.shape != (L, B, 2) or enc.dtype != torch.float32:
                raise RuntimeError("encode_mlp: enc must be the raw level-major lookup [L, B, 2] fp32 at these points")
        else:
            enc = torch.empty(L, B, 2, device=dev, dtype=torch.float32)
            call("pnr_grid_encode_forward", ptr(x01), ptr(emb), ptr(offsets), ptr(enc), _u32(B), _u32(3), _u32(2), _u32(L), _f32(S), _u32(H), None, _u32(gridtype),
                 _int(int(align_corners)), _int(0), units=B)
        t = 0 if tail is None else tail.shape[1]
        dims = [L * 2 + t] + [w.shape[0] for w in weights]
        desc = _desc(dims, act)
        lib = _lib.load()
        packed = torch.empty(int(lib.pnr_mlp_packed_bytes(ctypes.byref(desc))) // 4, dtype=torch.float32, device=dev)
        ws = [w.detach().contiguous() for w in weights]
        call("pnr_mlp_pack", ctypes.byref(desc), ptr(ws[0]), ptr(ws[1]), ptr(ws[2]) if len(ws) == 3 else None, ptr(packed))
        tail_c = None if tail is None else _al(tail.detach())
        y = torch.empty(B, dims[-1], dtype=torch.float32, device=dev)
        call("pnr_mlp_forward_lm", ctypes.byref(desc), ptr(packed), ptr(enc), ctypes.c_uint32(L), ptr(tail_c), ctypes.c_uint32(B), ptr(y))
        ctx.save_for_backward(x01, offsets, enc, tail_c, packed)
        ctx.dims, ctx.act, ctx.meta, ctx.rows = dims, act, meta, emb.shape[0]
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        from .gridencoder import _u32, _f32, _int
        x01, offsets, enc, tail_c, packed = ctx.saved_tensors
        dims, act = ctx.dims, ctx.act
        S, H, gridtype, align_corners = ctx.meta
        desc = _desc(dims, act)
        lib = _lib.load()
        dev = x01.device
        B, L = x01.shape[0], offsets.shape[0] - 1
        dy2 = _al(dy.reshape(-1, dims[-1]).float())
        want_emb = ctx.needs_input_grad[1]
        denc = torch.empty(L, B, 2, device=dev, dtype=torch.float32) if want_emb else None
        n = len(dims) - 1
        dws = [torch.empty(dims[l + 1], dims[l], dtype=torch.float32, device=dev) if ctx.needs_input_grad[7 + l] else None for l in range(n)]
        nbytes = int(lib.pnr_mlp_backward_workspace_bytes(ctypes.byref(desc), B))
        ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
        call("pnr_mlp_backward_lm", ctypes.byref(desc), ptr(packed), ptr(enc), ctypes.c_uint32(L), ptr(tail_c), ptr(dy2), ctypes.c_uint32(B), ptr(denc),
             ptr(dws[0]), ptr(dws[1]), ptr(dws[2]) if n == 3 else None, ptr(ws), ctypes.c_uint64(nbytes))
        grad_emb = None
        if want_emb:
            grad_emb = torch.zeros(ctx.rows, 2, device=dev, dtype=torch.float32)
            gb = int(lib.pnr_grid_backward_binned_workspace_bytes(B, L, ctx.rows))
            gws = torch.empty(gb // 4 + 1, dtype=torch.int32, device=dev)
            call("pnr_grid_encode_backward_binned", ptr(denc), ptr(x01), ptr(offsets), ptr(grad_emb), _u32(B), _u32(3), _u32(2), _u32(L), _f32(S), _u32(H),
                 _u32(gridtype), _int(int(align_corners)), ctypes.c_uint64(ctx.rows), ptr(gws), ctypes.c_uint64(gb))
        return (None, grad_emb, None, None, None, None, None, *dws)


def encode_mlp_fused_ok(encoder, x, tail, net, act=F.relu):
    """True when encode_mlp(encoder, x, ., tail, net, act) takes its fused path -- THE predicate: encode_mlp itself decides with this very function,
    so a caller that asks first (to hand over a lookup it already has) and encode_mlp cannot disagree."""
    from .gridencoder import BINNED_MIN_ROWS
    ok = (enabled and x.is_cuda and torch.is_grad_enabled() and not x.requires_grad and (tail is None or not tail.requires_grad)
          and getattr(encoder, "num_levels", 0) == 16 and getattr(encoder, "level_dim", 0) == 2 and getattr(encoder, "input_dim", 0) == 3
          and hasattr(encoder, "embeddings") and encoder.embeddings.dtype == torch.float32 and x.numel() // 3 >= max(MIN_ROWS, BINNED_MIN_ROWS)
          and x.numel() // 3 * 16 * 8 < 2 ** 32)
    if ok:
        ok = net[0].in_features == 32 + (0 if tail is None else tail.shape[-1]) and fusable(net, _Rows(x.numel() // 3, net[0].in_features, x), act)
    return bool(ok)


def encode_mlp(encoder, x, bound, tail, net, act=F.relu, enc=None):
    """`_run(net, cat([encoder(x, bound), tail]), act)` -- fused end to end when it can be (CUDA fp32 hash grid with 16 levels x 2 features,
    a large batch, no gradient wanted for x or tail), the plain composition otherwise.
    enc: the encoder's raw level-major output [16, B, 2] at x when the caller has it already -- a non-differentiable hint: the fused path reads it
    instead of looking the rows up again; when the fused path is refused it is ignored and the lookup recomputed (never an error in mid-step)."""
    import numpy as np
    if not encode_mlp_fused_ok(encoder, x, tail, net, act):
        h = encoder(x, bound=bound)
        if tail is not None:
            h = torch.cat([h, tail], dim=-1)
        return run_mlp(net, h, act)
    x01 = ((x + bound) / (2 * bound)).reshape(-1, 3)   # same two roundings as GridEncoder.forward (gridencoder/grid.py:142)
    meta = (float(np.log2(encoder.per_level_scale)), int(encoder.base_resolution), int(encoder.gridtype_id), bool(encoder.align_corners))
    t2 = None if tail is None else tail.reshape(-1, tail.shape[-1])
    y = _EncodeMLP.apply(x01, encoder.embeddings, encoder.offsets, meta, t2, _ACT[act], enc, *[l.weight for l in net])
    return y.reshape(*x.shape[:-1], y.shape[-1])


class _Rows:
    """Stand-in with the attributes `fusable` looks at (the [rows, in] input is never materialised on the fused path)."""

    def __init__(self, rows, width, like):
        self.is_cuda, self.dtype, self.shape, self.requires_grad = like.is_cuda, torch.float32, (rows, width), False
        self._n = rows * width

    def numel(self):
        return self._n
