"""Pure-PyTorch (CPU) hash-grid and SH encoders: the "pure-PyTorch path" SURVEY.md 8(d) names as the reference's CPU baseline for
BASELINE configs[0] (NeRFRenderer.run without the CUDA extensions).  TEST INFRASTRUCTURE: bench.py's cpu_baseline leg and tests/ only.

The reference itself ships no torch hash grid (its gridencoder is CUDA-only); this is the array statement of gridencoder.cu:35-175 with
torch ops -- int64 index arithmetic masked to 32 bits, fp32 interpolation in the reference's corner order -- and of shencoder.cu:50-68
(degree <= 4).  Checked against the C oracle in tests/test_oracle.py (grid: bit-exact up to fma contraction, i.e. <= 1 ulp-level; SH 1e-6).
"""
import numpy as np
import torch
import torch.nn as nn

from . import orc

_PRIMES = (1, 2654435761, 805459861)
_M32 = 0xFFFFFFFF


class TorchGridEncoder(nn.Module):
    """GridEncoder (gridencoder/grid.py:91-153; hash type, align_corners False, D = 3) with torch ops only."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=None,
                 gridtype="hash", align_corners=False):
        super().__init__()
        assert input_dim == 3 and gridtype == "hash" and not align_corners
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution = per_level_scale, base_resolution
        self.output_dim = num_levels * level_dim
        offs = orc.grid_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size)
        self.register_buffer("offsets", torch.from_numpy(offs))
        self.embeddings = nn.Parameter(torch.empty(int(offs[-1]), level_dim).uniform_(-1e-4, 1e-4))
        scale, res = orc.grid_level_params(num_levels, per_level_scale, base_resolution)   # the very numbers the kernels get (gridencoder.cu:125-126)
        self._scale, self._res = [float(s) for s in scale], [int(r) for r in res]

    def forward(self, inputs, bound=1):
        x = (inputs + bound) / (2 * bound)
        lead = list(x.shape[:-1])
        x = x.reshape(-1, 3)
        inside = ((x >= 0) & (x <= 1)).all(-1, keepdim=True)
        offs = self.offsets.tolist()
        outs = []
        for lv in range(self.num_levels):
            size, side = offs[lv + 1] - offs[lv], self._res[lv] + 1
            pos = torch.addcmul(torch.full_like(x, 0.5), x, torch.tensor(self._scale[lv]))     # fmaf(x, scale, 0.5) up to one rounding
            pg = torch.floor(pos)
            fr = pos - pg
            pg = pg.to(torch.int64).clamp_(min=0)
            dense = side ** 3 <= size
            acc = torch.zeros(x.shape[0], self.level_dim, dtype=x.dtype)
            table = self.embeddings[offs[lv]:offs[lv + 1]]
            for c in range(8):
                w = torch.ones(x.shape[0], dtype=x.dtype)
                idx3 = []
                for d in range(3):
                    bit = (c >> d) & 1
                    w = w * (fr[:, d] if bit else 1 - fr[:, d])
                    idx3.append(pg[:, d] + bit)
                if dense:
                    index = idx3[0] + idx3[1] * side + idx3[2] * (side * side)
                else:
                    index = ((idx3[0] * _PRIMES[0]) & _M32) ^ ((idx3[1] * _PRIMES[1]) & _M32) ^ ((idx3[2] * _PRIMES[2]) & _M32)
                index = index % size
                acc = acc + w[:, None] * table[index]
            outs.append(acc)
        out = torch.cat(outs, -1) * inside
        return out.view(lead + [self.output_dim])


class TorchSHEncoder(nn.Module):
    """SHEncoder (shencoder/sphere_harmonics.py:61-86), degree <= 4, constants of shencoder.cu:50-68."""

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        assert input_dim == 3 and 1 <= degree <= 4
        self.degree, self.output_dim = degree, degree ** 2

    def forward(self, inputs, size=1):
        v = inputs / size
        x, y, z = v[..., 0], v[..., 1], v[..., 2]
        xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
        o = [torch.full_like(x, 0.28209479177387814)]
        if self.degree > 1:
            o += [-0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x]
        if self.degree > 2:
            o += [1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999, -1.0925484305920792 * xz,
                  0.54627421529603959 * x2 - 0.54627421529603959 * y2]
        if self.degree > 3:
            o += [0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z, 0.45704579946446572 * y * (1.0 - 5.0 * z2),
                  0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
                  0.59004358992664352 * x * (-x2 + 3.0 * y2)]
        return torch.stack(o, -1)
