"""NumPy front-end of the C oracle (oracle/pnr_oracle.c).  TEST INFRASTRUCTURE ONLY.

Each wrapper allocates outputs exactly as the reference's Python layer does (zero-initialisation
contracts included: raymarching/raymarching.py:205-207,384-386; gridencoder/grid.py:72) and calls
the C restatement of the corresponding kernel.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liborc.so")
_VARIANTS = {"": "liborc.so", "omp": "liborc_omp.so", "nofma": "liborc_nofma.so"}   # see the header of pnr_oracle.c


def build(force=False):
    src = os.path.join(_HERE, "pnr_oracle.c")
    sos = [os.path.join(_HERE, "_build", f) for f in _VARIANTS.values()]
    if force or any(not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src) for so in sos):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_libs = {}
_variant = ""


def use_variant(name=""):
    """Select the build every wrapper below calls: "" canonical (single thread, explicit fmaf), "omp" (same arithmetic, all cores),
    "nofma" (no contraction anywhere: the reference's kernel bodies as g++ compiles them).  Returns the previous selection."""
    global _variant
    if name not in _VARIANTS:
        raise ValueError(name)
    prev, _variant = _variant, name
    return prev


def set_threads(n):
    """Thread count of the "omp" variant (no-op for the others)."""
    lib().orc_set_threads(ctypes.c_int(int(n)))


def lib():
    if _variant not in _libs:
        build()
        _libs[_variant] = ctypes.CDLL(os.path.join(_HERE, "_build", _VARIANTS[_variant]))
    return _libs[_variant]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


_u = ctypes.c_uint32
_f = ctypes.c_float
_i = ctypes.c_int


# ------------------------------------------------------------------ raymarching utils
def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    rays_o, rays_d, aabb = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3), _f32(aabb)
    N = rays_o.shape[0]
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    lib().orc_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), _u(N), _f(min_near), _p(nears), _p(fars))
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    coords = np.empty((N, 2), np.float32)
    lib().orc_sph_from_ray(_p(rays_o), _p(rays_d), _f(radius), _u(N), _p(coords))
    return coords


def morton3D(coords):
    coords = _i32(coords)
    N = coords.shape[0]
    out = np.empty(N, np.int32)
    lib().orc_morton3d(_p(coords), _u(N), _p(out))
    return out


def morton3D_invert(indices):
    indices = _i32(indices)
    N = indices.shape[0]
    out = np.empty((N, 3), np.int32)
    lib().orc_morton3d_invert(_p(indices), _u(N), _p(out))
    return out


def packbits(grid, thresh, bitfield=None):
    grid = _f32(grid)
    N = grid.size // 8
    if bitfield is None:
        bitfield = np.empty(N, np.uint8)
    lib().orc_packbits(_p(grid), _u(N), _f(thresh), _p(bitfield))
    return bitfield


# ------------------------------------------------------------------ training march / composite
def march_rays_train(rays_o, rays_d, bound, bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                     noises=None, align=-1, force_all_rays=False, dt_gamma=0.0, max_steps=1024):
    """raymarching/raymarching.py:161-235 (noises passed explicitly instead of torch.rand)."""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    M = N * max_steps
    if not force_all_rays and mean_count > 0:
        if align > 0:
            mean_count += align - mean_count % align
        M = mean_count
    xyzs, dirs = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    rays = np.empty((N, 3), np.int32)
    if step_counter is None:
        step_counter = np.zeros(2, np.int32)
    if noises is None:
        noises = np.zeros(N, np.float32)
    noises = _f32(noises)
    nears, fars = _f32(nears), _f32(fars)
    bitfield = np.ascontiguousarray(bitfield, dtype=np.uint8)
    lib().orc_march_rays_train(_p(rays_o), _p(rays_d), _p(bitfield), _f(bound), _f(dt_gamma), _u(max_steps), _u(N), _u(C),
                               _u(H), _u(M), _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(rays),
                               _p(step_counter), _p(noises))
    if force_all_rays or mean_count <= 0:
        m = int(step_counter[0])
        if align > 0:
            m += align - m % align
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    return xyzs, dirs, deltas, rays


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
    sigmas, rgbs, deltas, rays = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    ws, depth, image = np.empty(N, np.float32), np.empty(N, np.float32), np.empty((N, 3), np.float32)
    lib().orc_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), _u(M), _u(N), _f(T_thresh),
                                           _p(ws), _p(depth), _p(image))
    return ws, depth, image


def composite_rays_train_backward(grad_ws, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, T_thresh=1e-4):
    a = [_f32(v) for v in (grad_ws, grad_image, sigmas, rgbs, deltas)]
    rays, weights_sum, image = _i32(rays), _f32(weights_sum), _f32(image)
    M, N = a[2].shape[0], rays.shape[0]
    gs, gc = np.zeros_like(a[2]), np.zeros_like(a[3])
    lib().orc_composite_rays_train_backward(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(rays), _p(weights_sum),
                                            _p(image), _u(M), _u(N), _f(T_thresh), _p(gs), _p(gc))
    return gs, gc


def composite_rays_flex_train_forward(sigmas, inp, deltas, rays, T_thresh=1e-4):
    sigmas, inp, deltas, rays = _f32(sigmas), _f32(inp), _f32(deltas), _i32(rays)
    M, N, nc = sigmas.shape[0], rays.shape[0], inp.shape[-1]
    out = np.empty((N, nc), np.float32)
    lib().orc_composite_rays_flex_train_forward(_p(sigmas), _p(inp), _p(deltas), _p(rays), _u(M), _u(N), _u(nc),
                                                _f(T_thresh), _p(out))
    return out


def composite_rays_flex_train_backward(grad_out, sigmas, inp, deltas, rays, T_thresh=1e-4):
    grad_out, sigmas, inp, deltas, rays = _f32(grad_out), _f32(sigmas), _f32(inp), _f32(deltas), _i32(rays)
    M, N, nc = sigmas.shape[0], rays.shape[0], inp.shape[-1]
    gin = np.zeros_like(inp)
    lib().orc_composite_rays_flex_train_backward(_p(grad_out), _p(sigmas), _p(deltas), _p(rays), _u(M), _u(N), _u(nc),
                                                 _f(T_thresh), _p(gin))
    return gin


def spread_ray_to_sample(inp, rays, output):
    inp, rays = _f32(inp), _i32(rays)
    assert output.dtype == np.float32 and output.flags.c_contiguous
    lib().orc_spread_ray_to_sample(_p(inp), _p(rays), _u(output.shape[0]), _u(inp.shape[0]), _u(inp.shape[-1]), _p(output))


# ------------------------------------------------------------------ inference march / composite
def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield, C, H, nears, fars, align=-1,
               noises=None, dt_gamma=0.0, max_steps=1024):
    """raymarching/raymarching.py:347-398."""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)  # always pads (quirk 4)
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    if noises is None:
        noises = np.zeros(n_alive, np.float32)
    noises = _f32(noises)
    rays_alive, rays_t, nears, fars = _i32(rays_alive), _f32(rays_t), _f32(nears), _f32(fars)
    bitfield = np.ascontiguousarray(bitfield, dtype=np.uint8)
    lib().orc_march_rays(_u(n_alive), _u(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d), _f(bound),
                         _f(dt_gamma), _u(max_steps), _u(C), _u(H), _p(bitfield), _p(nears), _p(fars), _p(xyzs), _p(dirs),
                         _p(deltas), _p(noises))
    return xyzs, dirs, deltas


def _inplace(a, dt):
    assert a.dtype == dt and a.flags.c_contiguous, "in-place oracle ops need contiguous arrays of the exact dtype"
    return a


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """In-place on rays_alive, rays_t, weights_sum, depth, image (raymarching.py:401-423)."""
    sigmas, rgbs, deltas = _f32(sigmas), _f32(rgbs), _f32(deltas)
    lib().orc_composite_rays(_u(n_alive), _u(n_step), _f(T_thresh), _p(_inplace(rays_alive, np.int32)),
                             _p(_inplace(rays_t, np.float32)), _p(sigmas), _p(rgbs), _p(deltas),
                             _p(_inplace(weights_sum, np.float32)), _p(_inplace(depth, np.float32)),
                             _p(_inplace(image, np.float32)))


def composite_rays_flex(n_alive, n_step, n_channel, rays_alive, rays_t, sigmas, inp, deltas, weights_sum, output, T_thresh=1e-2):
    sigmas, inp, deltas = _f32(sigmas), _f32(inp), _f32(deltas)
    lib().orc_composite_rays_flex(_u(n_alive), _u(n_step), _u(n_channel), _f(T_thresh), _p(_inplace(rays_alive, np.int32)),
                                  _p(_inplace(rays_t, np.float32)), _p(sigmas), _p(inp), _p(deltas),
                                  _p(_inplace(weights_sum, np.float32)), _p(_inplace(output, np.float32)))


# ------------------------------------------------------------------ hash grid
def grid_offsets(input_dim=3, num_levels=16, per_level_scale=2.0, base_resolution=16, log2_hashmap_size=19, align_corners=False):
    """gridencoder/grid.py:111-121."""
    offsets, offset = [], 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(max_params, (resolution if align_corners else resolution + 1) ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offsets.append(offset)
        offset += n
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32)


def grid_level_params(L, per_level_scale, base_resolution):
    """Per-level (scale, resolution) exactly as the kernels receive them (gridencoder.cu:125-126)."""
    scale, res = np.empty(L, np.float32), np.empty(L, np.uint32)
    lib().orc_grid_level_params(_u(L), _f(np.float32(np.log2(per_level_scale))), _u(base_resolution), _p(scale), _p(res))
    return scale, res


def grid_encode_forward(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                        align_corners=False, raw=False):
    """gridencoder/grid.py:19-58.  Returns [B, L*C] (and dy_dx) unless raw=True ([L,B,C])."""
    inputs, offsets = _f32(inputs), _i32(offsets)
    B, D = inputs.shape
    L, C = offsets.shape[0] - 1, embeddings.shape[1]
    S = np.float32(np.log2(per_level_scale))
    if embeddings.dtype == np.float16:
        assert not calc_grad_inputs
        emb = np.ascontiguousarray(embeddings).view(np.uint16)
        out = np.empty((L, B, C), np.uint16)
        lib().orc_grid_encode_forward_half(_p(inputs), _p(emb), _p(offsets), _p(out), _u(B), _u(D), _u(C), _u(L), _f(S),
                                           _u(base_resolution), _u(gridtype), _i(int(align_corners)))
        out = out.view(np.float16)
        dy_dx = None
    else:
        emb = _f32(embeddings)
        out = np.empty((L, B, C), np.float32)
        dy_dx = np.empty((B, L * D * C), np.float32) if calc_grad_inputs else None
        lib().orc_grid_encode_forward(_p(inputs), _p(emb), _p(offsets), _p(out), _u(B), _u(D), _u(C), _u(L), _f(S),
                                      _u(base_resolution), _p(dy_dx), _u(gridtype), _i(int(align_corners)))
    if not raw:
        out = np.ascontiguousarray(out.transpose(1, 0, 2)).reshape(B, L * C)
    return (out, dy_dx) if calc_grad_inputs else out


def grid_encode_backward(grad, inputs, embeddings_shape, offsets, per_level_scale, base_resolution, gridtype=0,
                         align_corners=False, dy_dx=None):
    """gridencoder/grid.py:60-84.  grad: [B, L*C]."""
    inputs, offsets = _f32(inputs), _i32(offsets)
    B, D = inputs.shape
    L, C = offsets.shape[0] - 1, embeddings_shape[1]
    S = np.float32(np.log2(per_level_scale))
    g = np.ascontiguousarray(_f32(grad).reshape(B, L, C).transpose(1, 0, 2))
    gg = np.zeros(embeddings_shape, np.float32)
    lib().orc_grid_encode_backward(_p(g), _p(inputs), _p(offsets), _p(gg), _u(B), _u(D), _u(C), _u(L), _f(S),
                                   _u(base_resolution), _u(gridtype), _i(int(align_corners)))
    if dy_dx is None:
        return gg
    gi = np.zeros((B, D), np.float32)
    lib().orc_grid_input_backward(_p(g), _p(_f32(dy_dx)), _p(gi), _u(B), _u(D), _u(C), _u(L))
    return gg, gi


# ------------------------------------------------------------------ SH
def sh_encode_forward(inputs, degree, calc_grad_inputs=False):
    inputs = _f32(inputs)
    B, D = inputs.shape
    out = np.empty((B, degree * degree), np.float32)
    dy_dx = np.empty((B, D * degree * degree), np.float32) if calc_grad_inputs else None
    lib().orc_sh_encode_forward(_p(inputs), _p(out), _u(B), _u(D), _u(degree), _p(dy_dx))
    return (out, dy_dx) if calc_grad_inputs else out


def sh_encode_backward(grad, degree, dy_dx, D=3):
    grad, dy_dx = _f32(grad), _f32(dy_dx)
    B = grad.shape[0]
    gi = np.zeros((B, D), np.float32)
    lib().orc_sh_encode_backward(_p(grad), _u(B), _u(D), _u(degree), _p(dy_dx), _p(gi))
    return gi


# ------------------------------------------------------------------ palette
def rgb_to_hsv(x):
    x = _f32(x)
    shp = x.shape
    x = x.reshape(-1, 3)
    out = np.empty_like(x)
    lib().orc_rgb_to_hsv(_u(x.shape[0]), _p(x), _p(out))
    return out.reshape(shp)


def hsv_to_rgb(x):
    x = _f32(x)
    shp = x.shape
    x = x.reshape(-1, 3)
    out = np.empty_like(x)
    lib().orc_hsv_to_rgb(_u(x.shape[0]), _p(x), _p(out))
    return out.reshape(shp)


def get_rays(poses, intrinsics, H, W, inds=None):
    """nerf/utils.py:53-149, deterministic core: poses [B,4,4], intrinsics (fx,fy,cx,cy), inds [B,N] int64 or None (all pixels)."""
    poses = _f32(poses).reshape(-1, 4, 4)
    B = poses.shape[0]
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    if inds is not None:
        inds = np.ascontiguousarray(np.broadcast_to(np.asarray(inds, dtype=np.int64), (B, np.asarray(inds).shape[-1])))
    N = H * W if inds is None else inds.shape[1]
    rays_o, rays_d = np.empty((B, N, 3), np.float32), np.empty((B, N, 3), np.float32)
    lib().orc_get_rays(_p(poses), _u(B), _f(fx), _f(fy), _f(cx), _f(cy), _u(H), _u(W), _p(inds), _u(N), _p(rays_o), _p(rays_d))
    return rays_o, rays_d


def compute_RGB_histogram(colors_rgb, weights, bits_per_channel):
    colors_rgb, weights = _f32(colors_rgb), _f32(weights)
    nb = 1 << (3 * bits_per_channel)
    bw, bc = np.empty(nb, np.float64), np.empty((nb, 3), np.float32)
    lib().orc_rgb_histogram(_p(colors_rgb), _p(weights), _u(colors_rgb.shape[0]), _i(bits_per_channel), _p(bw), _p(bc))
    return bw, bc


def linear(x, W, bias=None):
    x, W = _f32(x), _f32(W)
    B, K = x.shape
    O = W.shape[0]
    y = np.empty((B, O), np.float32)
    b = None if bias is None else _f32(bias)
    lib().orc_linear(_p(x), _p(W), _p(b), _p(y), _u(B), _u(K), _u(O))
    return y


# ------------------------------------------------------------------ fields (nerf/network.py:95-124)
def nerf_field_forward(enc, dirs, w_sigma0, w_sigma1, w_color0, w_color1, w_color2):
    """enc: [B,32] hash-grid features, dirs: [B,3].  Returns (sigma [B], rgb [B,3]) -- every Linear is a
    k-ordered fp32 fmaf chain (orc_linear), activations in fp32."""
    h = np.maximum(linear(enc, w_sigma0), 0)
    h = linear(h, w_sigma1)
    sigma = np.exp(h[:, 0].astype(np.float32))
    geo = h[:, 1:]
    c = np.concatenate([sh_encode_forward(dirs, 4), geo], axis=1)
    c = np.maximum(linear(c, w_color0), 0)
    c = np.maximum(linear(c, w_color1), 0)
    o = linear(c, w_color2)
    rgb = (np.float32(1) / (np.float32(1) + np.exp(-o))).astype(np.float32)
    return sigma, rgb


# ------------------------------------------------------------------ occupancy maintenance (nerf/renderer.py:395-561)
def occupancy_points(C, H, bound, noise, coords=None, occ_rand=None, density_grid=None, n_partial=0, first=0, count=None):
    """Jittered world points of an occupancy sweep + the global cell id of each ([count,3] float32, [count] int32; id -1 = no sample).
    n_partial == 0: every cell once (Morton order); otherwise the partial sweep of renderer.py:512-537 with the caller's random numbers."""
    mode = 1 if n_partial else 0
    total = C * H ** 3 if mode == 0 else C * 2 * n_partial
    count = total - first if count is None else count
    pts = np.empty((count, 4), np.float32)
    noise = _f32(noise)
    coords = _i32(coords) if coords is not None else None
    occ_rand = _i32(occ_rand) if occ_rand is not None else None
    grid = _f32(density_grid) if density_grid is not None else None
    lib().orc_occupancy_points(_u(C), _u(H), _f(bound), _i(mode), _u(n_partial), _p(noise), _p(coords), _p(occ_rand), _p(grid), _u(first), _u(count), _p(pts))
    return pts[:, :3].copy(), pts[:, 3].copy().view(np.int32), pts


def occupancy_commit(density_grid, points4, candidates, decay, density_thresh):
    """EMA-max of `candidates` (sigma * density_scale per sample of `points4`) into density_grid (modified in place, [C, H^3] float32),
    then mean / threshold / packbits.  Returns (bitfield, mean, threshold)."""
    assert density_grid.dtype == np.float32 and density_grid.flags.c_contiguous
    C, H3 = density_grid.shape
    H = round(H3 ** (1 / 3))
    bits = np.empty(C * H3 // 8, np.uint8)
    state = np.zeros(2, np.float32)
    points4, candidates = _f32(points4), _f32(candidates)
    lib().orc_occupancy_commit(_u(C), _u(H), _p(density_grid), _p(points4), _p(candidates), _u(points4.shape[0]), _f(decay), _f(density_thresh), _p(bits), _p(state))
    return bits, float(state[0]), float(state[1])


def mark_untrained_grid(poses, intrinsic, density_grid, bound, min_near, filter_close_point=False):
    """nerf/renderer.py:395-465 on density_grid [C, H^3] (in place).  Returns the number of cells marked -1."""
    assert density_grid.dtype == np.float32 and density_grid.flags.c_contiguous
    C, H3 = density_grid.shape
    H = round(H3 ** (1 / 3))
    poses = _f32(poses).reshape(-1, 4, 4)
    fx, fy, cx, cy = [float(v) for v in intrinsic]
    n = np.zeros(1, np.int32)
    lib().orc_mark_untrained_grid(_p(poses), _u(poses.shape[0]), _f(fx), _f(fy), _f(cx), _f(cy), _u(C), _u(H), _f(bound), _f(min_near), _i(int(filter_close_point)),
                                  _p(density_grid), _p(n))
    return int(n[0])
