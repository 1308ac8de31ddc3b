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

from ._torch_glue import call, ptr, require

_u32, _f32, _int = ctypes.c_uint32, ctypes.c_float, ctypes.c_int

_gridtype_to_id = {"hash": 0, "tiled": 1}
_DTYPE_ID = {torch.float32: 0, torch.float16: 1}


class _grid_encode(Function):
    """gridencoder/grid.py:19-84"""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0, align_corners=False):
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = np.log2(per_level_scale)
        H = base_resolution
        # manual autocast handling, as the reference (grid.py:36-39): only the table goes to half
        if torch.is_autocast_enabled() and C % 2 == 0:
            embeddings = embeddings.to(torch.half)
        if embeddings.dtype not in _DTYPE_ID:
            raise RuntimeError("embeddings must be a float32 or float16 tensor")
        embeddings = embeddings.contiguous()
        outputs = torch.empty(L, B, C, device=inputs.device, dtype=embeddings.dtype)
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype) if calc_grad_inputs else None
        call("pnr_grid_encode_forward", ptr(require(inputs, torch.float32, "inputs")), ptr(require(embeddings, embeddings.dtype, "embeddings")),
             ptr(require(offsets, torch.int32, "offsets")), ptr(outputs), _u32(B), _u32(D), _u32(C), _u32(L), _f32(S), _u32(H), ptr(dy_dx),
             _u32(gridtype), _int(int(align_corners)), _int(_DTYPE_ID[embeddings.dtype]), units=B)
        outputs = outputs.permute(1, 0, 2).reshape(B, L * C)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, C, L, S, H, gridtype]
        ctx.align_corners = align_corners
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype = ctx.dims
        align_corners = ctx.align_corners
        grad = grad.view(B, L, C).permute(1, 0, 2).contiguous().to(embeddings.dtype)
        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if dy_dx is not None else None
        call("pnr_grid_encode_backward", ptr(grad), ptr(inputs), ptr(embeddings), ptr(offsets), ptr(grad_embeddings), _u32(B), _u32(D),
             _u32(C), _u32(L), _f32(S), _u32(H), ptr(dy_dx), ptr(grad_inputs), _u32(gridtype), _int(int(align_corners)),
             _int(_DTYPE_ID[embeddings.dtype]))
        if dy_dx is not None:
            grad_inputs = grad_inputs.to(inputs.dtype)
        return grad_inputs, grad_embeddings, None, None, None, None, None, None


grid_encode = _grid_encode.apply


class GridEncoder(nn.Module):
    """gridencoder/grid.py:91-153 -- same constructor, attributes, parameter/buffer names and init."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=4, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype="hash", align_corners=False):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.align_corners = align_corners

        offsets, offset = [], 0
        self.max_params = 2 ** log2_hashmap_size
        for i in range(num_levels):
            resolution = int(np.ceil(base_resolution * per_level_scale ** i))
            params_in_level = min(self.max_params, (resolution if align_corners else resolution + 1) ** input_dim)
            params_in_level = int(np.ceil(params_in_level / 8) * 8)
            offsets.append(offset)
            offset += params_in_level
        offsets.append(offset)
        self.register_buffer("offsets", torch.from_numpy(np.array(offsets, dtype=np.int32)))
        self.n_params = offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(offset, level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        std = 1e-4
        self.embeddings.data.uniform_(-std, std)

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} gridtype={self.gridtype} "
                f"align_corners={self.align_corners}")

    def forward(self, inputs, bound=1):
        inputs = (inputs + bound) / (2 * bound)
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        outputs = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, inputs.requires_grad,
                              self.gridtype_id, self.align_corners)
        return outputs.view(prefix_shape + [self.output_dim])
