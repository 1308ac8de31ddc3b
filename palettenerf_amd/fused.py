"""MI355X-first fused evaluation of the fields (no reference counterpart as one call): level-major
hash-grid lookup feeding the MFMA tiny-MLP kernel directly, no permute/copy, no per-layer launches."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._torch_glue import call, ptr, require

_u32, _f32, _int = ctypes.c_uint32, ctypes.c_float, ctypes.c_int


def grid_encode_raw(encoder, x01):
    """[B,3] in [0,1] -> raw level-major [L,B,C] fp32 encoder output (the layout the kernels produce)."""
    B = x01.shape[0]
    L, C = encoder.num_levels, encoder.level_dim
    out = torch.empty(L, B, C, device=x01.device, dtype=torch.float32)
    emb = encoder.embeddings.detach()
    call("pnr_grid_encode_forward", ptr(require(x01, torch.float32, "inputs")), ptr(require(emb, torch.float32, "embeddings")),
         ptr(encoder.offsets), ptr(out), _u32(B), _u32(encoder.input_dim), _u32(C), _u32(L), _f32(np.log2(encoder.per_level_scale)),
         _u32(encoder.base_resolution), None, _u32(encoder.gridtype_id), _int(int(encoder.align_corners)), _int(0), units=B)
    return out


class NeRFFieldFused:
    """Caches the MFMA-ordered weight blob of a NeRFNetwork and evaluates (sigma, rgb) for sample batches."""

    def __init__(self, model):
        self.model = model
        self.packed = None
        self.versions = None
        m = model
        ok = (m.encoder.num_levels == 16 and m.encoder.level_dim == 2 and m.encoder.input_dim == 3 and m.hidden_dim == 64 and m.geo_feat_dim == 15
              and m.num_layers == 2 and m.num_layers_color == 3 and m.hidden_dim_color == 64 and m.encoder_dir.degree == 4)
        if not ok:
            raise RuntimeError("fused NeRF field kernel is specialised for the shipped architecture (hashgrid 16x2, 64-wide nets, SH degree 4)")

    def _weights(self):
        m = self.model
        return [m.sigma_net[0].weight, m.sigma_net[1].weight, m.color_net[0].weight, m.color_net[1].weight, m.color_net[2].weight]

    def _pack(self):
        ws = self._weights()
        versions = tuple((w.data_ptr(), w._version) for w in ws)
        if self.packed is None or versions != self.versions:
            dev = ws[0].device
            if self.packed is None or self.packed.device != dev:
                self.packed = torch.empty(int(_lib.load().pnr_nerf_field_packed_bytes()) // 4, dtype=torch.float32, device=dev)
            ws = [require(w.detach().contiguous(), torch.float32, "weight") for w in ws]
            call("pnr_nerf_field_pack", *[ptr(w) for w in ws], ptr(self.packed))
            self.versions = versions
        return self.packed

    @torch.no_grad()
    def __call__(self, x, d):
        m = self.model
        x01 = ((x + m.bound) / (2 * m.bound)).contiguous()  # same two roundings as GridEncoder.forward (gridencoder/grid.py:142)
        enc = grid_encode_raw(m.encoder, x01)
        B = x.shape[0]
        sigmas = torch.empty(B, dtype=torch.float32, device=x.device)
        rgbs = torch.empty(B, 3, dtype=torch.float32, device=x.device)
        call("pnr_nerf_field_forward", ptr(enc), ptr(require(d.contiguous(), torch.float32, "dirs")), ptr(self._pack()), _u32(B), ptr(sigmas),
             ptr(rgbs), units=B)
        return sigmas, rgbs
