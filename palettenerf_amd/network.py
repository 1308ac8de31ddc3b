"""Fields evaluated per sample on the hot path: `NeRFNetwork` (nerf/network.py:10-143) and
`PaletteNetwork` (palette/network.py:10-280).  Same module/parameter names as the reference so a
reference checkpoint's state_dict loads (encoder.embeddings, sigma_net.N.weight, color_net.N.weight,
diff_net, basis_net, offsets_radiance_net, omega_net.0, encoder_palette, encoder_clip, clip_net,
basis_color, density_grid, density_bitfield, aabb_*, step_counter).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .activation import trunc_exp
from .encoding import get_encoder
from .linear import Linear
from .mlp import MIN_ROWS as _MLP_MIN_ROWS, encode_mlp, encode_mlp_fused_ok, run_mlp
from .palette_utils import palette_heads
from .renderer import NeRFRenderer, PaletteRenderer
from .shencoder import sh_encode_cat, sigma_geo_cat


PAIR_LOOKUP = True      # PaletteNeRF training: `encoder` (frozen density) and `encoder_palette` looked up in one launch (fused.grid_encode_raw_pair)


class _Probe:
    """What encode_mlp_fused_ok looks at in a tail tensor that does not exist yet (diffuse.detach(): [B, 3], no gradient)."""

    def __init__(self, like, width):
        self.requires_grad, self.shape = False, (like.shape[0], width)


def pairable(a, b):
    from .fused import pairable as f
    return f(a, b)


def grid_encode_raw_pair(a, b, x01):
    from .fused import grid_encode_raw_pair as f
    return f(a, b, x01)


def _mlp(dims):
    """Bias-free Linear stack (nerf/network.py:33-47); activations are applied by the caller."""
    return nn.ModuleList([Linear(dims[i], dims[i + 1], bias=False) for i in range(len(dims) - 1)])


def _run(net, h, act=F.relu, out=None):
    """The reference's layer loop (+ `out`: torch.sigmoid behind a colour head); on the GPU under autograd the whole stack is one fused launch
    each way (mlp.run_mlp)."""
    return run_mlp(net, h, act, out)


def density_fused(model):
    from .fused import density_fused as get
    return get(model)


def _fused_arch_ok(m):
    """The fused density kernel is specialised for the shipped architecture (hashgrid 16 x 2, 64-wide sigma_net, 15 geometry features)."""
    e = m.encoder
    return (getattr(e, "num_levels", 0) == 16 and getattr(e, "level_dim", 0) == 2 and m.hidden_dim == 64 and m.geo_feat_dim == 15 and m.num_layers == 2
            and m.num_layers_color == 3 and m.hidden_dim_color == 64 and getattr(m.encoder_dir, "degree", 0) == 4)


def _fused_density_ok(m, x):
    return bool(m.fused_field) and x.is_cuda and not torch.is_grad_enabled() and not torch.is_autocast_enabled() and _fused_arch_ok(m)


def _fused_heads_ok(m, h):
    """Training batches (fp32, autograd on, not under autocast) take the two colour heads through pnr_palette_heads_*."""
    return (h.is_cuda and h.ndim == 2 and h.dtype == torch.float32 and torch.is_grad_enabled() and not torch.is_autocast_enabled()
            and h.shape[0] >= _MLP_MIN_ROWS and h.shape[1] <= 16 and m.num_basis <= 10 and m.offsets_radiance_net.weight.dtype == torch.float32
            and (h.requires_grad or m.offsets_radiance_net.weight.requires_grad or m.omega_net[0].weight.requires_grad))


class NeRFNetwork(NeRFRenderer):
    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", num_layers=2, hidden_dim=64, geo_feat_dim=15,
                 num_layers_color=3, hidden_dim_color=64, bound=1, **kwargs):
        super().__init__(bound, **kwargs)
        if self.bg_radius > 0:
            raise NotImplementedError("background model (bg_radius > 0) is out of scope: every shipped scene config sets bg_radius=0")
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound)
        self.sigma_net = _mlp([self.in_dim] + [hidden_dim] * (num_layers - 1) + [1 + geo_feat_dim])
        self.encoder_dir, self.in_dim_dir = get_encoder(encoding_dir)
        self.color_net = _mlp([self.in_dim_dir + geo_feat_dim] + [hidden_dim_color] * (num_layers_color - 1) + [3])
        self.bg_net = None
        self.fused_field = False  # True: inference batches go through the fused MFMA field kernel (fp32, no autograd)
        self._fused = None

    def forward(self, x, d):
        """x: [N,3] in [-bound,bound]; d: [N,3] unit.  Returns (sigma [N], rgb [N,3]).  nerf/network.py:95-124"""
        if self.fused_field and not torch.is_grad_enabled() and not torch.is_autocast_enabled():
            if self._fused is None:
                from .fused import NeRFFieldFused
                self._fused = NeRFFieldFused(self)
            return self._fused(x, d)
        h = encode_mlp(self.encoder, x, self.bound, None, self.sigma_net)
        sigma, cat = sigma_geo_cat(self.encoder_dir, h, d)      # trunc_exp(h[..., 0]); cat([encoder_dir(d), h[..., 1:]]): one launch each way
        return sigma, _run(self.color_net, cat, out=torch.sigmoid)

    def density(self, x):
        """nerf/network.py:126-143"""
        if _fused_density_ok(self, x):
            sigma, geo = density_fused(self)(x)
            return {"sigma": sigma, "geo_feat": geo}
        h = encode_mlp(self.encoder, x, self.bound, None, self.sigma_net)
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    density._pnr_fused_density = True   # exp(sigma_net(encoder(x))): what pnr_occupancy_update evaluates itself (renderer._fused_sweep_ok)

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        """nerf/network.py:157-184 -- colour head alone, optionally only where mask is set (other rows stay 0)."""
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
            if not mask.any():
                return rgbs
            d, geo_feat = d[mask], geo_feat[mask]
        h = _run(self.color_net, sh_encode_cat(self.encoder_dir, d, geo_feat), out=torch.sigmoid)
        if mask is None:
            return h
        rgbs[mask] = h.to(rgbs.dtype)
        return rgbs

    def get_params(self, lr):
        """nerf/network.py:186-206"""
        return [{"params": self.encoder.parameters(), "lr": lr}, {"params": self.sigma_net.parameters(), "lr": lr},
                {"params": self.encoder_dir.parameters(), "lr": lr}, {"params": self.color_net.parameters(), "lr": lr}]


class PaletteNetwork(PaletteRenderer):
    def __init__(self, opt, encoding="hashgrid", encoding_dir="sphere_harmonics", num_layers=2, hidden_dim=64, geo_feat_dim=15,
                 num_layers_color=3, hidden_dim_color=64, bound=1, **kwargs):
        super().__init__(opt, bound, **kwargs)
        if self.bg_radius > 0:
            raise NotImplementedError("background model (bg_radius > 0) is out of scope: every shipped scene config sets bg_radius=0")
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color
        self.encoder, self.in_dim = get_encoder(encoding, desired_resolution=2048 * bound)
        self.encoder_palette, self.in_dim_palette = get_encoder(encoding, desired_resolution=2048 * bound)
        self.encoder_clip, self.in_dim_clip = get_encoder(encoding, desired_resolution=2048 * bound)
        self.num_basis = opt.num_basis
        self.sigma_net = _mlp([self.in_dim] + [hidden_dim] * (num_layers - 1) + [1 + geo_feat_dim])
        self.encoder_dir, self.in_dim_dir = get_encoder(encoding_dir)
        # named color_net so that the vanilla NeRF checkpoint's colour head loads (palette/network.py:58-59)
        self.color_net = _mlp([self.in_dim_dir + geo_feat_dim] + [hidden_dim] * (num_layers_color - 1) + [3])
        self.diff_net = _mlp([geo_feat_dim] + [hidden_dim] * (num_layers_color - 1) + [3])
        self.basis_net = _mlp([self.in_dim_palette + 3] + [hidden_dim] * (num_layers - 1) + [geo_feat_dim])
        self.offsets_radiance_net = Linear(geo_feat_dim, self.num_basis * 3 + 1)  # the only layer with a bias
        self.omega_net = nn.Sequential(Linear(geo_feat_dim, self.num_basis, bias=False), nn.Softplus())
        if opt.pred_clip:
            self.clip_net = _mlp([self.in_dim_clip] + [hidden_dim] * (num_layers - 1) + [opt.clip_dim])
        self.bg_net = None
        self.fused_field = False  # True: inference goes through the fused palette field + packed-aux composite (no edit / stylizer)
        self._fused = None

    def forward(self, x, d, frozen_density=False):
        """palette/network.py:156-185.  Returns sigma, clip_feat, omega, offsets_radiance, view_dep, diffuse.
        frozen_density: the caller detaches sigma (PaletteNeRF training, palette/renderer.py:333-334; geo_feat is detached here anyway), so
        encoder + sigma_net may run as the fused no-gradient density kernel."""
        enc_pal = None
        if frozen_density and x.is_cuda and _fused_arch_ok(self):   # also under autocast: fp32 tables and MFMA chains, no gradient needed
            enc_density = None
            if PAIR_LOOKUP and x.ndim == 2 and not torch.is_autocast_enabled() and pairable(self.encoder, self.encoder_palette) \
                    and encode_mlp_fused_ok(self.encoder_palette, x, _Probe(x, 3), self.basis_net, F.elu):
                # both tables are read at the same points (palette/network.py:161, 257): one pair lookup instead of two (fused.grid_encode_raw_pair)
                x01 = ((x + self.bound) / (2 * self.bound)).contiguous()
                enc_density, enc_pal = grid_encode_raw_pair(self.encoder, self.encoder_palette, x01)
            sigma, geo_feat = density_fused(self)(x, enc=enc_density)
        else:
            h = encode_mlp(self.encoder, x, self.bound, None, self.sigma_net)
            sigma = trunc_exp(h[..., 0])
            geo_feat = h[..., 1:].detach()
        if self.opt.pred_clip:
            clip_feat = encode_mlp(self.encoder_clip, x, self.bound, None, self.clip_net)
        else:
            if self.training and torch.is_grad_enabled():   # a read-only broadcast of one zero (nobody writes the training batch's clip_feat): no fill of [M, clip_dim]
                z = getattr(self, "_zero", None)
                if z is None or z.device != sigma.device or z.dtype != sigma.dtype:
                    z = self._zero = torch.zeros(1, dtype=sigma.dtype, device=sigma.device)
                clip_feat = z.expand(*sigma.shape, self.opt.clip_dim)
            else:
                clip_feat = sigma.new_zeros(*sigma.shape, self.opt.clip_dim)   # palette/network.py:179 (zeros_like of a repeat there)
        omega, offsets_radiance, view_dep, diffuse = self.color(x, d, geo_feat=geo_feat, enc_palette=enc_pal)
        return sigma, clip_feat, omega, offsets_radiance, view_dep, diffuse

    def density(self, x):
        if _fused_density_ok(self, x):
            sigma, geo = density_fused(self)(x)
            return {"sigma": sigma, "geo_feat": geo}
        h = encode_mlp(self.encoder, x, self.bound, None, self.sigma_net)
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    density._pnr_fused_density = True

    def color(self, x, d, mask=None, geo_feat=None, enc_palette=None, **kwargs):
        """palette/network.py:223-280 (unmasked form: the march path never passes a mask).
        enc_palette: encoder_palette's raw level-major output at x when forward() has looked it up together with the density table."""
        if mask is not None:
            raise NotImplementedError("masked colour queries belong to the non-cuda_ray path, which is dead code in the reference")
        g = geo_feat.detach()
        diffuse = _run(self.diff_net, g, out=torch.sigmoid)
        view_dep = _run(self.color_net, sh_encode_cat(self.encoder_dir, d, g), out=torch.sigmoid)
        h = encode_mlp(self.encoder_palette, x, self.bound, diffuse.detach(), self.basis_net, act=F.elu, enc=enc_palette)   # cat([encoder_palette(x), diffuse]) -> basis_net
        if _fused_heads_ok(self, h):
            offsets_radiance, omega = palette_heads(h, self.offsets_radiance_net, self.omega_net[0])
        else:
            offsets_radiance = self.offsets_radiance_net(h)
            omega = self.omega_net(h) + 0.05
            omega = omega / omega.sum(dim=-1, keepdim=True)
        return omega, offsets_radiance, view_dep, diffuse

    def get_params(self, lr):
        """palette/network.py:283-308 -- basis_net (and clip_net unless pred_clip) are deliberately absent (quirk 8)."""
        params = [{"params": m.parameters(), "lr": lr} for m in
                  (self.encoder, self.encoder_palette, self.encoder_clip, self.sigma_net, self.encoder_dir, self.color_net, self.diff_net,
                   self.offsets_radiance_net, self.omega_net)]
        params.append({"params": self.basis_color, "lr": lr})
        if self.opt.use_initialization_from_rgbxy and hasattr(self, "hist_weights"):
            params.append({"params": self.hist_weights, "lr": lr})
        if self.opt.pred_clip:
            params.append({"params": self.clip_net.parameters(), "lr": lr})
        return params
