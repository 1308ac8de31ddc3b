"""The reference's three pybind11 extension modules (`_raymarching`, `_gridencoder`, `_shencoder`) as CPU stand-ins over the C oracle.
TEST INFRASTRUCTURE (used by tests/golden/gen_golden.py and tests/test_oracle.py only).

Why: the reference's operator wrappers -- gridencoder/grid.py:19-153, shencoder/sphere_harmonics.py:14-86 and the composite
Functions of raymarching/raymarching.py:238-341,401-473 -- contain no forced `.cuda()`, and each tries `import _<name> as _backend`
before it falls back to a JIT build (raymarching.py:9-12, grid.py:9-12, sphere_harmonics.py:9-12).  With these modules seeded into
`sys.modules` the reference's OWN wrappers import and run on the CPU, so the offsets, `per_level_scale`, the [L,B,C] buffer and its
permute, the half-table cast under autocast, the `(x + bound) / (2 bound)` map, the zero-initialisation contracts and the autograd
plumbing in the fixtures come from the reference itself; only the kernel bodies are the oracle's.

Every function has the argument list of the reference's C++ declaration (cited), takes CPU torch tensors, and writes its outputs in
place through their storage -- exactly the contract of the pybind layer (raw data_ptr, no copies, contiguity is the caller's business:
checked here, because a stand-in that silently copied would hide a stride bug the CUDA kernels would not forgive).
"""
import ctypes
import types

import torch

from . import orc

_u, _f, _i = ctypes.c_uint32, ctypes.c_float, ctypes.c_int


def _p(t, dtype=None):
    if t is None:
        return None
    assert isinstance(t, torch.Tensor) and not t.is_cuda, "the CPU stand-in takes CPU tensors"
    assert t.is_contiguous(), "the reference's kernels read raw data_ptr(): tensors must be contiguous"
    if dtype is not None:
        assert t.dtype == dtype, (t.dtype, dtype)
    return ctypes.c_void_p(t.data_ptr())


F32, I32, U8, F16 = torch.float32, torch.int32, torch.uint8, torch.float16


def make_native_backends():
    lib = orc.lib
    rm = types.ModuleType("_raymarching")   # raymarching/src/raymarching.h:7-23

    def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
        lib().orc_near_far_from_aabb(_p(rays_o, F32), _p(rays_d, F32), _p(aabb, F32), _u(N), _f(min_near), _p(nears, F32), _p(fars, F32))

    def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
        lib().orc_composite_rays_train_forward(_p(sigmas, F32), _p(rgbs, F32), _p(deltas, F32), _p(rays, I32), _u(M), _u(N), _f(T_thresh), _p(weights_sum, F32),
                                               _p(depth, F32), _p(image, F32))

    def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs):
        lib().orc_composite_rays_train_backward(_p(grad_weights_sum, F32), _p(grad_image, F32), _p(sigmas, F32), _p(rgbs, F32), _p(deltas, F32), _p(rays, I32),
                                                _p(weights_sum, F32), _p(image, F32), _u(M), _u(N), _f(T_thresh), _p(grad_sigmas, F32), _p(grad_rgbs, F32))

    def composite_rays_flex_train_forward(sigmas, input, deltas, rays, M, N, n_channel, T_thresh, output):
        lib().orc_composite_rays_flex_train_forward(_p(sigmas, F32), _p(input, F32), _p(deltas, F32), _p(rays, I32), _u(M), _u(N), _u(n_channel), _f(T_thresh),
                                                    _p(output, F32))

    def composite_rays_flex_train_backward(grad_output, sigmas, input, deltas, rays, output, M, N, n_channel, T_thresh, grad_input):
        # (the kernel reads neither `input` nor `output`: raymarching.cu:764-819)
        lib().orc_composite_rays_flex_train_backward(_p(grad_output, F32), _p(sigmas, F32), _p(deltas, F32), _p(rays, I32), _u(M), _u(N), _u(n_channel), _f(T_thresh),
                                                     _p(grad_input, F32))

    def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        lib().orc_composite_rays(_u(n_alive), _u(n_step), _f(T_thresh), _p(rays_alive, I32), _p(rays_t, F32), _p(sigmas, F32), _p(rgbs, F32), _p(deltas, F32),
                                 _p(weights_sum, F32), _p(depth, F32), _p(image, F32))

    def composite_rays_flex(n_alive, n_step, n_channel, T_thresh, rays_alive, rays_t, sigmas, input, deltas, weights, output):
        lib().orc_composite_rays_flex(_u(n_alive), _u(n_step), _u(n_channel), _f(T_thresh), _p(rays_alive, I32), _p(rays_t, F32), _p(sigmas, F32), _p(input, F32),
                                      _p(deltas, F32), _p(weights, F32), _p(output, F32))

    def spread_ray_to_sample(input, rays, M, N, n_channel, output):
        lib().orc_spread_ray_to_sample(_p(input, F32), _p(rays, I32), _u(M), _u(N), _u(n_channel), _p(output, F32))

    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs, dirs, deltas, noises):
        lib().orc_march_rays(_u(n_alive), _u(n_step), _p(rays_alive, I32), _p(rays_t, F32), _p(rays_o, F32), _p(rays_d, F32), _f(bound), _f(dt_gamma), _u(max_steps),
                             _u(C), _u(H), _p(grid, U8), _p(nears, F32), _p(fars, F32), _p(xyzs, F32), _p(dirs, F32), _p(deltas, F32), _p(noises, F32))

    for fn in (near_far_from_aabb, composite_rays_train_forward, composite_rays_train_backward, composite_rays_flex_train_forward,
               composite_rays_flex_train_backward, composite_rays, composite_rays_flex, spread_ray_to_sample, march_rays):
        setattr(rm, fn.__name__, fn)

    ge = types.ModuleType("_gridencoder")   # gridencoder/src/gridencoder.h:12-13

    def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners):
        if embeddings.dtype == F16:
            assert dy_dx is None, "the oracle's half path has no dy_dx"
            lib().orc_grid_encode_forward_half(_p(inputs, F32), _p(embeddings, F16), _p(offsets, I32), _p(outputs, F16), _u(B), _u(D), _u(C), _u(L), _f(S), _u(H),
                                               _u(gridtype), _i(int(align_corners)))
        else:
            lib().orc_grid_encode_forward(_p(inputs, F32), _p(embeddings, F32), _p(offsets, I32), _p(outputs, F32), _u(B), _u(D), _u(C), _u(L), _f(S), _u(H),
                                          _p(dy_dx, F32), _u(gridtype), _i(int(align_corners)))

    def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners):
        lib().orc_grid_encode_backward(_p(grad, F32), _p(inputs, F32), _p(offsets, I32), _p(grad_embeddings, F32), _u(B), _u(D), _u(C), _u(L), _f(S), _u(H),
                                       _u(gridtype), _i(int(align_corners)))
        if dy_dx is not None:
            lib().orc_grid_input_backward(_p(grad, F32), _p(dy_dx, F32), _p(grad_inputs, F32), _u(B), _u(D), _u(C), _u(L))

    ge.grid_encode_forward, ge.grid_encode_backward = grid_encode_forward, grid_encode_backward

    sh = types.ModuleType("_shencoder")   # shencoder/src/shencoder.h:9-10

    def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
        lib().orc_sh_encode_forward(_p(inputs, F32), _p(outputs, F32), _u(B), _u(D), _u(C), _p(dy_dx, F32))

    def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
        lib().orc_sh_encode_backward(_p(grad, F32), _u(B), _u(D), _u(C), _p(dy_dx, F32), _p(grad_inputs, F32))

    sh.sh_encode_forward, sh.sh_encode_backward = sh_encode_forward, sh_encode_backward
    return rm, ge, sh
