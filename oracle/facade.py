"""torch-CPU facades over the NumPy oracle, shaped like the reference's extension modules
(`raymarching`, `gridencoder`, `shencoder`, `palette.utils`).  TEST INFRASTRUCTURE: used by
tests/golden/gen_golden.py (to drive the reference's Python callers) and by the CPU tests that run
this repo's host-side renderer/network mirror without a GPU."""
import types

import numpy as np
import torch

from . import orc as oracle


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_oracle_modules():
    """torch-CPU facades over the NumPy oracle, shaped like the reference's extension modules."""
    rm = types.ModuleType("raymarching")

    def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
        n, f = oracle.near_far_from_aabb(rays_o.numpy(), rays_d.numpy(), aabb.numpy(), min_near)
        return _t(n), _t(f)

    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield, C, H, near, far, align=-1, perturb=False,
                   dt_gamma=0, max_steps=1024):
        assert not perturb
        x, d, dl = oracle.march_rays(n_alive, n_step, rays_alive.numpy(), rays_t.numpy(), rays_o.numpy(), rays_d.numpy(), bound,
                                     bitfield.numpy(), C, H, near.numpy(), far.numpy(), align, None, dt_gamma, max_steps)
        return _t(x), _t(d), _t(dl)

    def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
        oracle.composite_rays(n_alive, n_step, rays_alive.numpy(), rays_t.numpy(), sigmas.detach().numpy(), rgbs.detach().numpy(),
                              deltas.numpy(), weights_sum.numpy(), depth.numpy(), image.numpy(), T_thresh)
        return tuple()

    def composite_rays_flex(n_alive, n_step, n_channel, rays_alive, rays_t, sigmas, inp, deltas, weights_sum, output, T_thresh=1e-2):
        oracle.composite_rays_flex(n_alive, n_step, n_channel, rays_alive.numpy(), rays_t.numpy(), sigmas.detach().numpy(),
                                   inp.detach().contiguous().numpy(), deltas.numpy(), weights_sum.numpy(), output.numpy(), T_thresh)
        return tuple()

    def march_rays_train(rays_o, rays_d, bound, bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False, align=-1,
                         force_all_rays=False, dt_gamma=0, max_steps=1024):
        assert not perturb
        cnt = step_counter.numpy()
        x, d, dl, r = oracle.march_rays_train(rays_o.numpy(), rays_d.numpy(), bound, bitfield.numpy(), C, H, nears.numpy(), fars.numpy(),
                                              cnt, mean_count, None, align, force_all_rays, dt_gamma, max_steps)
        return _t(x), _t(d), _t(dl), _t(r)

    class _CompositeTrain(torch.autograd.Function):
        @staticmethod
        def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
            ws, dp, im = oracle.composite_rays_train_forward(sigmas.numpy(), rgbs.numpy(), deltas.numpy(), rays.numpy(), T_thresh)
            ws, dp, im = _t(ws), _t(dp), _t(im)
            ctx.save_for_backward(sigmas, rgbs, deltas, rays, ws, im)
            ctx.T = T_thresh
            return ws, dp, im

        @staticmethod
        def backward(ctx, gws, gdp, gim):
            sigmas, rgbs, deltas, rays, ws, im = ctx.saved_tensors
            gs, gc = oracle.composite_rays_train_backward(gws.contiguous().numpy(), gim.contiguous().numpy(), sigmas.numpy(), rgbs.numpy(),
                                                          deltas.numpy(), rays.numpy(), ws.numpy(), im.numpy(), ctx.T)
            return _t(gs), _t(gc), None, None, None

    class _CompositeFlexTrain(torch.autograd.Function):
        @staticmethod
        def forward(ctx, sigmas, inp, deltas, rays, T_thresh=1e-4):
            inp = inp.contiguous()
            out = _t(oracle.composite_rays_flex_train_forward(sigmas.numpy(), inp.numpy(), deltas.numpy(), rays.numpy(), T_thresh))
            ctx.save_for_backward(sigmas, inp, deltas, rays)
            ctx.T = T_thresh
            return out

        @staticmethod
        def backward(ctx, go):
            sigmas, inp, deltas, rays = ctx.saved_tensors
            gi = oracle.composite_rays_flex_train_backward(go.contiguous().numpy(), sigmas.numpy(), inp.numpy(), deltas.numpy(), rays.numpy(), ctx.T)
            return None, _t(gi), None, None, None

    def spread_ray_to_sample(inp, rays, output):
        oracle.spread_ray_to_sample(inp.numpy(), rays.numpy(), output.numpy())
        return tuple()

    rm.near_far_from_aabb, rm.march_rays, rm.composite_rays, rm.composite_rays_flex = near_far_from_aabb, march_rays, composite_rays, composite_rays_flex
    rm.march_rays_train, rm.spread_ray_to_sample = march_rays_train, spread_ray_to_sample
    rm.composite_rays_train = lambda s, c, d, r, T=1e-4: _CompositeTrain.apply(s.contiguous(), c.contiguous(), d, r, T)
    rm.composite_rays_flex_train = lambda s, i, d, r, T=1e-4: _CompositeFlexTrain.apply(s.contiguous(), i, d, r, T)
    rm.morton3D = lambda c: _t(oracle.morton3D(c.numpy()))
    rm.morton3D_invert = lambda i: _t(oracle.morton3D_invert(i.numpy()))
    rm.packbits = lambda g, t, b=None: _t(oracle.packbits(g.numpy(), t))

    # ---- gridencoder facade: same GridEncoder class body as the product's (parameter names), oracle compute
    ge = types.ModuleType("gridencoder")

    class _GridFn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution):
            out = oracle.grid_encode_forward(inputs.numpy(), embeddings.detach().numpy(), offsets.numpy(), per_level_scale, base_resolution)
            ctx.save_for_backward(inputs, offsets)
            ctx.meta = (tuple(embeddings.shape), per_level_scale, base_resolution)
            return _t(out)

        @staticmethod
        def backward(ctx, grad):
            inputs, offsets = ctx.saved_tensors
            shp, pls, br = ctx.meta
            gg = oracle.grid_encode_backward(grad.contiguous().numpy(), inputs.numpy(), shp, offsets.numpy(), pls, br)
            return None, _t(gg), None, None, None

    class GridEncoder(torch.nn.Module):
        def __init__(self, input_dim=3, num_levels=16, level_dim=4, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                     desired_resolution=None, gridtype="hash", align_corners=False):
            super().__init__()
            assert gridtype == "hash" and not align_corners
            if desired_resolution is not None:
                per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
            self.input_dim, self.per_level_scale, self.base_resolution = input_dim, per_level_scale, base_resolution
            self.output_dim = num_levels * level_dim
            offs = oracle.grid_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size)
            self.register_buffer("offsets", torch.from_numpy(offs))
            self.embeddings = torch.nn.Parameter(torch.empty(int(offs[-1]), level_dim).uniform_(-1e-4, 1e-4))

        def forward(self, inputs, bound=1):
            inputs = (inputs + bound) / (2 * bound)
            prefix = list(inputs.shape[:-1])
            out = _GridFn.apply(inputs.reshape(-1, self.input_dim).contiguous(), self.embeddings, self.offsets, self.per_level_scale,
                                self.base_resolution)
            return out.view(prefix + [self.output_dim])

    ge.GridEncoder = GridEncoder

    sh = types.ModuleType("shencoder")

    class SHEncoder(torch.nn.Module):
        def __init__(self, input_dim=3, degree=4):
            super().__init__()
            self.degree, self.output_dim = degree, degree ** 2

        def forward(self, inputs, size=1):
            inputs = inputs / size
            prefix = list(inputs.shape[:-1])
            out = oracle.sh_encode_forward(inputs.reshape(-1, 3).contiguous().numpy(), self.degree)
            return _t(out).reshape(prefix + [self.output_dim])

    sh.SHEncoder = SHEncoder

    pu = types.ModuleType("palette.utils")
    pu.normalize = lambda t: t / (t.norm(dim=-1, keepdim=True) + 1e-9)
    pu.rgb_to_hsv = lambda t: _t(oracle.rgb_to_hsv(t.detach().contiguous().numpy()))
    pu.hsv_to_rgb = lambda t: _t(oracle.hsv_to_rgb(t.detach().contiguous().numpy()))
    return rm, ge, sh, pu
