"""The ray-level tail of a training step as one HIP launch each way (csrc/train_loss.hip, `pnr_train_loss_*` in include/pnr.h).

A training-mode `run_cuda` returns a `TrainResults` dict: the composited maps are there as before, while `image`, `depth` and `direct_rgb`
(palette/renderer.py:387-403, nerf/renderer.py:328-332: background blend and depth normalisation, ~12 small torch launches and as many in
the backward) are formed only when somebody reads them.  `train_loss(results, gt_rgb, ...)` is the trainer's loss (palette/utils.py:483-600
with the MSE criterion of main_palette.py:222; nerf/utils.py:534-556) computed straight from the raw composites `results.raw` -- blend, depth,
every loss term, and in the backward every gradient of weights_sum / image / all_map -- without touching those lazy entries.  Same value as the
torch formulation to rounding (sums are reduced in a fixed order of their own); any other criterion keeps using the dict entries.
"""
import collections
import ctypes

import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from . import _lib
from ._torch_glue import ptr, require, stream_ptr

RawTrain = collections.namedtuple("RawTrain", "weights_sum depth_raw image_raw all_map nears fars bg_color prefix num_basis clip_dim")
TERM_NAMES = ("loss", "loss_mse", "loss_sparsity", "loss_offsets", "loss_view_dep", "loss_smooth", "loss_palette", "loss_weight", "loss_direct",
              "loss_clip_feat")   # terms[i]; the names of palette/utils.py:551-577's loss_dict


class TrainResults(dict):
    """dict of a training-mode render; `image`, `depth`, `direct_rgb` are computed (with autograd, the reference's formulas) on first access."""
    _LAZY = ("image", "depth", "direct_rgb")

    def __init__(self, raw, eager=()):
        super().__init__(eager)
        self.raw = raw

    def _has(self, key):
        return key in self._LAZY and (key != "direct_rgb" or self.raw.all_map is not None)

    def __missing__(self, key):
        if not self._has(key):
            raise KeyError(key)
        r = self.raw
        if key == "image":
            v = (r.image_raw + (1 - r.weights_sum).unsqueeze(-1) * r.bg_color).view(*r.prefix, 3)
        elif key == "depth":
            v = (torch.clamp(r.depth_raw - r.nears, min=0) / (r.fars - r.nears)).view(*r.prefix)
        else:
            v = (r.all_map[..., 7:10] + (1 - r.weights_sum).unsqueeze(-1) * r.bg_color).view(*r.prefix, 3)
        self[key] = v
        return v

    def __contains__(self, key):
        return super().__contains__(key) or self._has(key)

    def get(self, key, default=None):
        return self[key] if key in self else default


_workspaces = {}   # (device, stream) -> zero-initialised int32 tensor (the ticket stays zero between launches: the kernel resets it)


def _workspace(device, nbytes):
    """The last-workgroup ticket + partial-sum rows of pnr_train_loss_forward.  One buffer per (device, STREAM): launches on one stream are
    ordered, launches on different streams (or from different host threads, which use different streams or serialise on one) must not share
    the ticket.  (The kernel leaves the ticket at zero; a launch that faults before it does takes the HIP context with it, so there is no
    next launch to poison.)"""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.zeros((nbytes + 3) // 4 * 2, dtype=torch.int32, device=device)
        _workspaces[key] = ws
    return ws


def _rows(t, n, width, name):
    if t is None:
        return None
    t = t.detach().reshape(-1, width) if width else t.detach().reshape(-1)
    if t.shape[0] != n:
        raise RuntimeError(f"train_loss: {name} must have one row per ray")
    return require(t.float().contiguous(), torch.float32, name)


class _train_loss(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, weights_sum, image_raw, all_map, basis_color, cfg):
        dev, f32 = weights_sum.device, torch.float32
        N = weights_sum.shape[0]
        a = _lib.TrainLossArgs()
        keep = []

        def P(t):
            keep.append(t)
            return ptr(t)
        ws_ = require(weights_sum.detach().contiguous(), f32, "weights_sum")
        im_ = require(image_raw.detach().contiguous(), f32, "image")
        am_ = None if all_map is None else require(all_map.detach().contiguous(), f32, "all_map")
        a.N, a.num_basis, a.clip_dim = N, cfg["num_basis"], cfg["clip_dim"]
        a.n_channel = 0 if am_ is None else am_.shape[1]
        a.weights_sum, a.image_raw, a.all_map = P(ws_), P(im_), P(am_)
        want = cfg["want_outputs"]
        depth_raw = _rows(cfg["depth_raw"], N, 0, "depth") if want else None
        a.depth_raw, a.nears, a.fars = P(depth_raw), P(_rows(cfg["nears"], N, 0, "nears")), P(_rows(cfg["fars"], N, 0, "fars"))
        bg = cfg["bg_color"]
        if torch.is_tensor(bg):
            if bg.requires_grad:
                raise RuntimeError("train_loss: a background that needs a gradient (bg_radius > 0) takes the torch formulation")
            bg = bg.detach().to(dev, f32)
            if bg.numel() == 1:
                a.bg_mode, a.bg_const = 0, float(bg)
            elif bg.numel() == 3:
                a.bg_mode, a.bg_color = 1, P(bg.reshape(3).contiguous())
            else:
                a.bg_mode, a.bg_color = 2, P(_rows(bg, N, 3, "bg_color"))
        else:
            a.bg_mode, a.bg_const = 0, float(bg)
        a.gt_rgb = P(_rows(cfg["gt_rgb"], N, 3, "gt_rgb"))
        a.gt_clip = P(_rows(cfg["gt_clip"], N, a.clip_dim, "gt_clip")) if cfg["gt_clip"] is not None and am_ is not None else None
        a.gt_weights = P(_rows(cfg["gt_weights"], N, a.num_basis, "gt_weights")) if cfg["gt_weights"] is not None and am_ is not None else None
        bc = None if basis_color is None else require(basis_color.detach().contiguous(), f32, "basis_color")
        a.basis_color = P(bc)
        a.basis_color_origin = P(None if bc is None else require(cfg["basis_color_origin"].detach().to(dev, f32).contiguous(), f32, "basis_color_origin"))
        for k in ("lambda_sparsity", "lambda_offsets", "lambda_view_dep", "lambda_smooth", "lambda_weight", "lambda_palette"):
            setattr(a, k, float(cfg[k]))
        terms = torch.empty(_lib.TRAIN_LOSS_TERMS, device=dev, dtype=f32)
        loss_ray = torch.empty(N, device=dev, dtype=f32)
        image = torch.empty(N, 3, device=dev, dtype=f32) if want else None
        depth = torch.empty(N, device=dev, dtype=f32) if depth_raw is not None else None
        direct = torch.empty(N, 3, device=dev, dtype=f32) if want and am_ is not None else None
        a.image, a.depth, a.direct_rgb, a.loss_ray, a.terms = ptr(image), ptr(depth), ptr(direct), ptr(loss_ray), ptr(terms)
        nbytes = int(_lib.load().pnr_train_loss_workspace_bytes(N))
        wsp = _workspace(dev, nbytes)
        a.workspace, a.workspace_bytes = ptr(wsp), nbytes
        _lib.call("pnr_train_loss_forward", ctypes.byref(a), stream_ptr())
        ctx.args, ctx.keep = a, keep
        ctx.shapes = (weights_sum.shape, image_raw.shape, None if all_map is None else all_map.shape, None if basis_color is None else basis_color.shape)
        ctx.set_materialize_grads(False)   # five of the six outputs carry no gradient: no zero tensors are made for them
        loss = terms[0]
        outs = (loss, terms.detach(), loss_ray, image, depth, direct)
        ctx.mark_non_differentiable(*[o for o in outs[1:] if o is not None])
        return outs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g_loss, *_):
        if g_loss is None:
            return None, None, None, None, None
        a = ctx.args
        dev, f32 = g_loss.device, torch.float32
        N, C = a.N, a.n_channel
        g = g_loss.detach().reshape(1).float().contiguous()
        g_ws = torch.empty(N, device=dev, dtype=f32)
        g_im = torch.empty(N, 3, device=dev, dtype=f32)
        g_am = torch.empty(N, C, device=dev, dtype=f32) if C else None
        g_bc = torch.empty(a.num_basis, 3, device=dev, dtype=f32) if ctx.shapes[3] is not None and ctx.needs_input_grad[3] else None
        a.grad_loss, a.grad_weights_sum, a.grad_image_raw, a.grad_all_map, a.grad_basis_color = ptr(g), ptr(g_ws), ptr(g_im), ptr(g_am), ptr(g_bc)
        _lib.call("pnr_train_loss_backward", ctypes.byref(a), stream_ptr())
        s = ctx.shapes
        return (g_ws.view(s[0]), g_im.view(s[1]), None if g_am is None else g_am.view(s[2]), None if g_bc is None else g_bc.view(s[3]), None)


def train_loss(results, gt_rgb, lambda_sparsity=0.0, lambda_offsets=0.0, lambda_view_dep=0.0, lambda_smooth=0.0, lambda_weight=0.0,
               lambda_palette=0.0, gt_weights=None, gt_clip=None, basis_color=None, basis_color_origin=None, want_outputs=True):
    """palette/utils.py:483-600 (`train_step` from `pred_rgb = outputs['image']` to `loss = loss.mean()`) on a TrainResults, MSE criterion:
    returns (loss, info) with info = {"terms": [10] tensor in TERM_NAMES order, "loss_ray": [N] per-ray colour error (the error map's input
    before the scalar terms), "image"/"depth"/"direct_rgb": detached renders for logging (None with want_outputs=False)}.
    gt_weights: the palette-weight guide (`get_palette_weight_with_hist`), gt_clip: feature targets (both optional; a clip term needs the model's
    pred_clip head).  basis_color + basis_color_origin add the palette anchor term.  A NeRF model's results (no all_map) give the colour term only
    (nerf/utils.py:535); its `lambda_sparse` term stays with the caller."""
    raw = getattr(results, "raw", results)
    if not isinstance(raw, RawTrain):
        raise RuntimeError("train_loss needs the TrainResults of a training-mode run_cuda (results.raw)")
    if not raw.weights_sum.is_cuda:
        raise RuntimeError("train_loss: expected CUDA(HIP) tensors (no CPU fallback exists)")
    if (basis_color is None) != (basis_color_origin is None):
        raise RuntimeError("train_loss: basis_color and basis_color_origin come together")
    cfg = dict(num_basis=raw.num_basis, clip_dim=raw.clip_dim, depth_raw=raw.depth_raw, nears=raw.nears, fars=raw.fars, bg_color=raw.bg_color, gt_rgb=gt_rgb,
               gt_clip=gt_clip, gt_weights=gt_weights, basis_color_origin=basis_color_origin, lambda_sparsity=lambda_sparsity, lambda_offsets=lambda_offsets,
               lambda_view_dep=lambda_view_dep, lambda_smooth=lambda_smooth, lambda_weight=lambda_weight, lambda_palette=lambda_palette,
               want_outputs=want_outputs)
    loss, terms, loss_ray, image, depth, direct = _train_loss.apply(raw.weights_sum, raw.image_raw, raw.all_map, basis_color, cfg)
    shaped = lambda t, *tail: None if t is None else t.view(*raw.prefix, *tail)   # noqa: E731
    return loss, {"terms": terms, "loss_ray": shaped(loss_ray), "image": shaped(image, 3), "depth": shaped(depth), "direct_rgb": shaped(direct, 3)}
