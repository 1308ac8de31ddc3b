"""Multiresolution hash-grid encoder -- same API as the reference's gridencoder/grid.py
(`_grid_encode`, `grid_encode`, `GridEncoder` with parameter `embeddings` and buffer `offsets`, so
state_dicts round-trip), backed by the gfx950 kernels behind pnr_grid_encode_{forward,backward}.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from . import _lib
from ._torch_glue import call, ptr, require

_u32, _f32, _int = ctypes.c_uint32, ctypes.c_float, ctypes.c_int

_gridtype_to_id = {"hash": 0, "tiled": 1}
_DTYPE_ID = {torch.float32: 0, torch.float16: 1}
BINNED_MIN_ROWS = 32768   # batches from this size on use the bucket-binned table gradient
ROW_LAYOUT = False        # True: [B, L*C] rows straight from the kernel (pnr_grid_encode_forward_layout; same values).  Measured SLOWER than [L, B, C] + permute-copy
                          # (2^20 samples: 336 us against 176 + 70 us, profiles/r04_grid_op_bench.txt): 8-byte stores at a 128-byte stride are partial-line writes


def level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners=False):
    """Row offset of every level inside the shared table (same sizing rule as the reference, gridencoder/grid.py:111-121):
    a level holds min(2^log2_hashmap_size, (res [+1])^D) rows rounded up to a multiple of 8."""
    res = np.ceil(base_resolution * per_level_scale ** np.arange(num_levels)).astype(np.int64)
    side = res if align_corners else res + 1
    rows = np.minimum(2 ** log2_hashmap_size, side.astype(object) ** input_dim).astype(np.int64)
    rows = (rows + 7) // 8 * 8
    return np.concatenate([[0], np.cumsum(rows)]).astype(np.int32)


class _grid_encode(Function):
    """(inputs [B,D] in [0,1], embeddings [rows,C], offsets [L+1]) -> [B, L*C]; same positional signature as the reference's
    `_grid_encode` (gridencoder/grid.py:19-84).  The native kernels work level-major ([L,B,C])."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0, align_corners=False):
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L, C = offsets.shape[0] - 1, embeddings.shape[1]
        S, H = np.log2(per_level_scale), base_resolution
        if torch.is_autocast_enabled() and C % 2 == 0:
            embeddings = embeddings.to(torch.half)  # under autocast only the TABLE goes to half (gridencoder/grid.py:36-39)
        if embeddings.dtype not in _DTYPE_ID:
            raise RuntimeError("embeddings must be a float32 or float16 tensor")
        embeddings = embeddings.contiguous()
        table_dtype = embeddings.dtype
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=table_dtype) if calc_grad_inputs else None
        # D = 3, C = 2 without dy_dx (every shipped configuration): the kernel writes the [B, L*C] rows this function returns; otherwise the
        # reference's [L, B, C] and its permute-copy (gridencoder/grid.py:41,57)
        rows = ROW_LAYOUT and D == 3 and C == 2 and dy_dx is None
        out = torch.empty((B, L * C) if rows else (L, B, C), device=inputs.device, dtype=table_dtype)
        call("pnr_grid_encode_forward_layout", ptr(require(inputs, torch.float32, "inputs")), ptr(require(embeddings, table_dtype, "embeddings")),
             ptr(require(offsets, torch.int32, "offsets")), ptr(out), _u32(B), _u32(D), _u32(C), _u32(L), _f32(S), _u32(H), ptr(dy_dx),
             _u32(gridtype), _int(int(align_corners)), _int(_DTYPE_ID[table_dtype]), _int(1 if rows else 0), units=B)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.meta = (B, D, C, L, S, H, gridtype, bool(align_corners))
        if rows:
            return out
        return out.permute(1, 0, 2).reshape(B, L * C)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, align_corners = ctx.meta
        grad_level_major = grad.view(B, L, C).permute(1, 0, 2).contiguous().to(embeddings.dtype)
        grad_embeddings = torch.zeros_like(embeddings)  # the scatter accumulates into it
        if dy_dx is None and D == 3 and C == 2 and embeddings.dtype == torch.float32 and B >= BINNED_MIN_ROWS and B * L * 8 < 2 ** 32:
            # large training batches: bucket-binned accumulation in LDS instead of 16 L B float atomics (csrc/grid_binned.hip)
            rows = embeddings.shape[0]
            nbytes = int(_lib.load().pnr_grid_backward_binned_workspace_bytes(B, L, rows))
            ws = torch.empty(nbytes // 4 + 1, dtype=torch.int32, device=embeddings.device)
            call("pnr_grid_encode_backward_binned", ptr(grad_level_major), ptr(inputs), ptr(offsets), ptr(grad_embeddings), _u32(B), _u32(D), _u32(C),
                 _u32(L), _f32(S), _u32(H), _u32(gridtype), _int(int(align_corners)), ctypes.c_uint64(rows), ptr(ws), ctypes.c_uint64(nbytes))
            return None, grad_embeddings, None, None, None, None, None, None
        grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if dy_dx is not None else None
        call("pnr_grid_encode_backward", ptr(grad_level_major), ptr(inputs), ptr(embeddings), ptr(offsets), ptr(grad_embeddings), _u32(B), _u32(D),
             _u32(C), _u32(L), _f32(S), _u32(H), ptr(dy_dx), ptr(grad_inputs), _u32(gridtype), _int(int(align_corners)),
             _int(_DTYPE_ID[embeddings.dtype]))
        if grad_inputs is not None:
            grad_inputs = grad_inputs.to(inputs.dtype)
        return grad_inputs, grad_embeddings, None, None, None, None, None, None


grid_encode = _grid_encode.apply


class GridEncoder(nn.Module):
    """Multiresolution hash / tiled grid with the reference's constructor, attributes and state_dict layout
    (parameter `embeddings` [rows, level_dim] ~ U(-1e-4, 1e-4), buffer `offsets` int32 [L+1]; gridencoder/grid.py:91-153)."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=4, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype="hash", align_corners=False):
        super().__init__()
        if desired_resolution is not None:  # geometric progression from base_resolution to desired_resolution
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution, self.log2_hashmap_size = per_level_scale, base_resolution, log2_hashmap_size
        self.output_dim = num_levels * level_dim
        self.gridtype, self.gridtype_id, self.align_corners = gridtype, _gridtype_to_id[gridtype], align_corners
        self.max_params = 2 ** log2_hashmap_size
        offsets = level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners)
        self.register_buffer("offsets", torch.from_numpy(offsets))
        self.n_params = int(offsets[-1]) * level_dim
        self.embeddings = nn.Parameter(torch.empty(int(offsets[-1]), level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        with torch.no_grad():   # in place on the Parameter itself (not `.data`): bumps its version, so blobs derived from the table are rebuilt
            self.embeddings.uniform_(-1e-4, 1e-4)

    def __repr__(self):
        finest = int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {finest} per_level_scale={self.per_level_scale:.4f} "
                f"params={tuple(self.embeddings.shape)} gridtype={self.gridtype} align_corners={self.align_corners}")

    def forward(self, inputs, bound=1):
        """inputs [..., input_dim] in [-bound, bound] -> [..., num_levels * level_dim]"""
        unit = (inputs + bound) / (2 * bound)
        lead = list(unit.shape[:-1])
        flat = unit.view(-1, self.input_dim)
        encoded = grid_encode(flat, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, flat.requires_grad,
                              self.gridtype_id, self.align_corners)
        return encoded.view(lead + [self.output_dim])
