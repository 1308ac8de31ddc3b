"""Hot-path pieces of the reference's palette/utils.py: the HSV operators used by RegionEdit inside
the inference loop (palette/utils.py:257-295) and `normalize` (:126-127)."""
import ctypes

import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from ._torch_glue import call, ptr, require, to_cuda

_u32 = ctypes.c_uint32


def normalize(tensor):
    return tensor / (tensor.norm(dim=-1, keepdim=True) + 1e-9)


class _rgb_to_hsv(Function):
    """palette/utils.py:258-274 (forward only)"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, input):
        input = to_cuda(input)
        prefix = input.shape[:-1]
        input = input.contiguous().view(-1, 3)
        n = input.shape[0]
        output = torch.empty(n, 3, device=input.device, dtype=input.dtype)
        call("pnr_rgb_to_hsv", _u32(n), ptr(require(input, torch.float32, "input")), ptr(output))
        return output.reshape(*prefix, 3)


rgb_to_hsv = _rgb_to_hsv.apply


class _hsv_to_rgb(Function):
    """palette/utils.py:278-293 (forward only)"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, input):
        input = to_cuda(input)
        prefix = input.shape[:-1]
        input = input.contiguous().view(-1, 3)
        n = input.shape[0]
        output = torch.empty(n, 3, device=input.device, dtype=input.dtype)
        call("pnr_hsv_to_rgb", _u32(n), ptr(require(input, torch.float32, "input")), ptr(output))
        return output.reshape(*prefix, 3)


hsv_to_rgb = _hsv_to_rgb.apply


def compute_RGB_histogram(colors_rgb, weights, bits_per_channel):
    """palette/utils.py:129-146: (colors_rgb [n,3] f32, weights [n] f32, bits) -> (bin_weights f64 [2^(3b)], bin_centers f32 [2^(3b),3]).
    NumPy in / NumPy out like the reference (whose implementation is CPU C++); torch tensors on the GPU are accepted and returned as is."""
    import numpy as np
    as_numpy = isinstance(colors_rgb, np.ndarray)
    c = torch.as_tensor(colors_rgb, dtype=torch.float32)
    w = torch.as_tensor(weights, dtype=torch.float32)
    assert c.ndim == 2 and c.shape[1] == 3 and w.ndim == 1 and len(c) == len(w) and 1 <= bits_per_channel <= 8
    c, w = to_cuda(c).contiguous(), to_cuda(w).contiguous()
    nb = 1 << (3 * bits_per_channel)
    bw = torch.empty(nb, dtype=torch.float64, device=c.device)
    bc = torch.empty(nb, 3, dtype=torch.float32, device=c.device)
    call("pnr_rgb_histogram", ptr(c), ptr(w), _u32(c.shape[0]), ctypes.c_int(bits_per_channel), ptr(bw), ptr(bc))
    return (bw.cpu().numpy(), bc.cpu().numpy()) if as_numpy else (bw, bc)


class _palette_train_shade(Function):
    """Training-mode palette colour-basis composite (palette/renderer.py:344-386) as one HIP launch each way: see
    `pnr_palette_train_shade_forward` in include/pnr.h.  Returns (rgbs [M,3], all_buffer [M, 13 + clip_dim + nb])."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, omega, offsets_radiance, view_dep, diffuse, clip_feat, smooth_norm, basis_color, clip_dim=None):
        M, nb = omega.shape
        if clip_dim is None:
            clip_dim = 0 if clip_feat is None else clip_feat.shape[-1]
        if clip_feat is not None and clip_feat.shape != (M, clip_dim):
            raise RuntimeError("palette_train_shade: clip_feat must be [M, clip_dim]")
        f32 = torch.float32
        omega, offsets_radiance = require(omega.contiguous(), f32, "omega"), require(offsets_radiance.contiguous(), f32, "offsets_radiance")
        view_dep, diffuse = require(view_dep.contiguous(), f32, "view_dep"), require(diffuse.contiguous(), f32, "diffuse")
        basis_color = require(basis_color.contiguous(), f32, "basis_color")
        if offsets_radiance.shape != (M, 3 * nb + 1) or basis_color.shape != (nb, 3):
            raise RuntimeError("palette_train_shade: offsets_radiance must be [M, 3 nb + 1] and basis_color [nb, 3]")
        if clip_feat is not None:
            clip_feat = require(clip_feat.contiguous(), f32, "clip_feat")
        if smooth_norm is not None:
            smooth_norm = require(smooth_norm.contiguous().view(-1), f32, "smooth_norm")
        rgbs = torch.empty(M, 3, device=omega.device, dtype=f32)
        all_buffer = torch.empty(M, 13 + clip_dim + nb, device=omega.device, dtype=f32)
        call("pnr_palette_train_shade_forward", _u32(M), _u32(nb), _u32(clip_dim), ptr(omega), ptr(offsets_radiance), ptr(view_dep), ptr(diffuse),
             ptr(clip_feat), ptr(smooth_norm), ptr(basis_color), ptr(rgbs), ptr(all_buffer))
        ctx.save_for_backward(omega, offsets_radiance, view_dep, basis_color)
        ctx.dims = (M, nb, clip_dim, clip_feat is not None, smooth_norm is not None)
        return rgbs, all_buffer

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g_rgbs, g_all):
        from . import _lib
        omega, offsets_radiance, view_dep, basis_color = ctx.saved_tensors
        M, nb, clip_dim, has_clip, has_smooth = ctx.dims
        dev, f32 = omega.device, torch.float32
        g_rgbs = torch.zeros(M, 3, device=dev, dtype=f32) if g_rgbs is None else g_rgbs.contiguous().float()
        g_all = torch.zeros(M, 13 + clip_dim + nb, device=dev, dtype=f32) if g_all is None else g_all.contiguous().float()
        g_omega, g_or = torch.empty_like(omega), torch.empty_like(offsets_radiance)
        g_vd, g_df = torch.empty(M, 3, device=dev, dtype=f32), torch.empty(M, 3, device=dev, dtype=f32)
        g_clip = torch.empty(M, clip_dim, device=dev, dtype=f32) if has_clip and ctx.needs_input_grad[4] else None
        g_smooth = torch.empty(M, 1, device=dev, dtype=f32) if has_smooth and ctx.needs_input_grad[5] else None
        g_bc = ws = None
        ws_bytes = 0
        if ctx.needs_input_grad[6]:
            g_bc = torch.empty(nb, 3, device=dev, dtype=f32)
            ws_bytes = int(_lib.load().pnr_palette_train_shade_workspace_bytes(nb))
            ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        call("pnr_palette_train_shade_backward", _u32(M), _u32(nb), _u32(clip_dim), ptr(omega), ptr(offsets_radiance), ptr(view_dep), ptr(basis_color),
             ptr(g_rgbs), ptr(g_all), ptr(g_omega), ptr(g_or), ptr(g_vd), ptr(g_df), ptr(g_clip), ptr(g_smooth), ptr(g_bc), ptr(ws),
             ctypes.c_uint64(ws_bytes))
        return g_omega, g_or, g_vd, g_df, g_clip, g_smooth, g_bc, None


palette_train_shade = _palette_train_shade.apply


class _palette_heads(Function):
    """offsets_radiance_net + omega_net + normalisation of PaletteNetwork.color (palette/network.py:262-268) as one HIP launch each way:
    see `pnr_palette_heads_forward` in include/pnr.h.  h [M, in] -> (offsets_radiance [M, 3 nb + 1], omega [M, nb])."""

    @staticmethod
    def forward(ctx, h, w_or, b_or, w_om):
        f32 = torch.float32
        h, w_or = require(h.contiguous(), f32, "h"), require(w_or.contiguous(), f32, "offsets_radiance_net.weight")
        b_or, w_om = require(b_or.contiguous(), f32, "offsets_radiance_net.bias"), require(w_om.contiguous(), f32, "omega_net.0.weight")
        M, n_in = h.shape
        nb = w_om.shape[0]
        if w_or.shape != (3 * nb + 1, n_in) or b_or.shape != (3 * nb + 1,) or w_om.shape != (nb, n_in):
            raise RuntimeError("palette_heads: offsets_radiance_net must be Linear(in, 3 nb + 1) and omega_net.0 Linear(in, nb, bias=False)")
        offsets_radiance = torch.empty(M, 3 * nb + 1, device=h.device, dtype=f32)
        omega = torch.empty(M, nb, device=h.device, dtype=f32)
        call("pnr_palette_heads_forward", ptr(h), ptr(w_or), ptr(b_or), ptr(w_om), _u32(M), _u32(nb), _u32(n_in), ptr(offsets_radiance), ptr(omega))
        ctx.save_for_backward(h, w_or, w_om)
        return offsets_radiance, omega

    @staticmethod
    def backward(ctx, g_or, g_omega):
        from .linear import bias_grad, weight_grad
        h, w_or, w_om = ctx.saved_tensors
        M, n_in = h.shape
        nb = w_om.shape[0]
        orw = 3 * nb + 1
        dev, f32 = h.device, torch.float32
        g_or = torch.zeros(M, orw, device=dev, dtype=f32) if g_or is None else g_or.contiguous().float()
        g_omega = torch.zeros(M, nb, device=dev, dtype=f32) if g_omega is None else g_omega.contiguous().float()
        g_h = torch.empty_like(h) if ctx.needs_input_grad[0] else None
        g_pre = torch.empty(M, orw + nb, device=dev, dtype=f32)
        call("pnr_palette_heads_backward", ptr(h), ptr(w_or), ptr(w_om), ptr(g_or), ptr(g_omega), _u32(M), _u32(nb), _u32(n_in), ptr(g_h), ptr(g_pre))
        g_w_or = g_b = g_w_om = None
        if M == 0:
            return g_h, torch.zeros_like(w_or), torch.zeros(orw, device=dev, dtype=f32), torch.zeros_like(w_om)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[3]:
            g_w = weight_grad(h, g_pre)            # [3 nb + 1 + nb, in]: both heads in one reduction
            g_w_or, g_w_om = g_w[:orw], g_w[orw:]
        if ctx.needs_input_grad[2]:
            g_b = bias_grad(g_pre)[:orw]
        return g_h, g_w_or, g_b, g_w_om


def palette_heads(h, offsets_radiance_net, omega_linear):
    """(offsets_radiance, omega) from basis_net's output through the fused heads kernel."""
    return _palette_heads.apply(h, offsets_radiance_net.weight, offsets_radiance_net.bias, omega_linear.weight)


def get_palette_weight_with_hist(rgb, hist_weights):
    """palette/utils.py:117-124: per-pixel palette weights read from the extraction's 3-D colour LUT `hist_weights` [1, nb, R, G, B]
    (PaletteRenderer.initialize_palette keeps it in that layout) by trilinear interpolation at the pixel's colour; zero outside [0, 1]^3.
    rgb [..., 3] in [0, 1] -> [..., nb].  A torch op in the reference as well (grid_sample; its x axis is the LUT's last one, i.e. blue)."""
    if hist_weights.ndim != 5:
        raise AssertionError("hist_weights must be [1, nb, R, G, B]")
    lead = rgb.shape[:-1]
    grid = rgb.reshape(1, 1, 1, -1, 3).flip(-1) * 2 - 1
    w = torch.nn.functional.grid_sample(hist_weights, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    return w.reshape(hist_weights.shape[1], -1).permute(1, 0).reshape(*lead, -1)
