"""The reference's Python operator layer restated over the REFERENCE-BUILT extension modules (oracle/_ref/ref_*.so: raymarching.cu, shencoder.cu,
palette.cu of /root/reference compiled for gfx950, oracle/ref_build.py).  TEST INFRASTRUCTURE: tests/test_gpu_reference_kernels.py and bench.py's
`extra.reference_kernels` leg only.  The reference's raymarching/raymarching.py cannot travel to the GPU box, so the few lines each wrapper adds
around its `_backend` call -- which buffers it allocates, which are zero-filled, the always-pad alignment, the slice to the counter -- are restated
here, citing the lines; the kernels underneath are the reference's own."""
import types

import torch

from . import ref_build


def available():
    return all(ref_build.hip_available(e) for e in ("raymarching", "shencoder", "palette"))


def raymarching_module():
    """An object with the reference's `raymarching` API (raymarching/raymarching.py) over the reference's own kernels."""
    be = ref_build.load_hip("raymarching")
    m = types.ModuleType("raymarching_reference_kernels")
    m._backend = be

    def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):                                   # raymarching.py:19-49
        rays_o, rays_d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        nears, fars = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device), torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        be.near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars)
        return nears, fars

    def morton3D(coords):                                                                         # :83-104
        coords = coords.int().contiguous()
        out = torch.empty(coords.shape[0], dtype=torch.int32, device=coords.device)
        be.morton3D(coords, coords.shape[0], out)
        return out

    def morton3D_invert(indices):                                                                 # :106-126
        indices = indices.int().contiguous()
        out = torch.empty(indices.shape[0], 3, dtype=torch.int32, device=indices.device)
        be.morton3D_invert(indices, indices.shape[0], out)
        return out

    def packbits(grid, thresh, bitfield=None):                                                    # :129-155
        grid = grid.contiguous()
        N = grid.shape[0] * grid.shape[1] // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        be.packbits(grid, N, thresh, bitfield)
        return bitfield

    def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False, align=-1,
                         force_all_rays=False, dt_gamma=0, max_steps=1024):                        # :161-235
        rays_o, rays_d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        N, dev = rays_o.shape[0], rays_o.device
        M = N * max_steps
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        noises = torch.rand(N, device=dev) if perturb else torch.zeros(N, device=dev)
        be.march_rays_train(rays_o, rays_d, density_bitfield, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, step_counter, noises)
        if force_all_rays or mean_count <= 0:
            m_ = int(step_counter[0].item())
            if align > 0:
                m_ += align - m_ % align
            xyzs, dirs, deltas = xyzs[:m_], dirs[:m_], deltas[:m_]
        return xyzs, dirs, deltas, rays

    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1, perturb=False,
                   dt_gamma=0, max_steps=1024):                                                   # :347-398
        rays_o, rays_d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        dev = rays_o.device
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)
        xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
        noises = torch.rand(n_alive, device=dev) if perturb else torch.zeros(n_alive, device=dev)
        be.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, density_bitfield, near, far, xyzs, dirs, deltas, noises)
        return xyzs, dirs, deltas

    def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):       # :401-423
        be.composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas.float().contiguous(), rgbs.float().contiguous(), deltas, weights_sum, depth, image)
        return tuple()

    def composite_rays_flex(n_alive, n_step, n_channel, rays_alive, rays_t, sigmas, input, deltas, weights_sum, output, T_thresh=1e-2):   # :425-447
        be.composite_rays_flex(n_alive, n_step, n_channel, T_thresh, rays_alive, rays_t, sigmas.float().contiguous(), input.float().contiguous(), deltas, weights_sum, output)
        return tuple()

    for f in (near_far_from_aabb, morton3D, morton3D_invert, packbits, march_rays_train, march_rays, composite_rays, composite_rays_flex):
        setattr(m, f.__name__, f)
    return m


class SHEncoder(torch.nn.Module):
    """shencoder/sphere_harmonics.py:61-86 (forward only) over the reference's kernel_sh."""

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree, self.output_dim = input_dim, degree, degree ** 2
        self._be = ref_build.load_hip("shencoder")

    def forward(self, inputs, size=1):
        inputs = inputs / size
        lead = list(inputs.shape[:-1])
        flat = inputs.reshape(-1, self.input_dim).float().contiguous()
        out = torch.empty(flat.shape[0], self.output_dim, dtype=flat.dtype, device=flat.device)
        self._be.sh_encode_forward(flat, out, flat.shape[0], self.input_dim, self.degree, None)
        return out.reshape(lead + [self.output_dim])


class swapped_in:
    """Context manager: this repository's renderer / network mirror running on the REFERENCE's march, composite and SH kernels (the hash grid
    stays this repository's: gridencoder.cu does not compile for HIP).  `sh_encode_cat` -- this repository's fused [SH | geo] launch -- is
    replaced by the reference's plain `torch.cat([encoder_dir(d), geo_feat])` (nerf/network.py:109-115) so that the SH values really come
    from kernel_sh."""

    def __enter__(self):
        from palettenerf_amd import network, renderer
        import palettenerf_amd.shencoder as psh
        self._mods = (renderer, psh, network)
        self._saved = (renderer.raymarching, psh.SHEncoder, network.sh_encode_cat)
        renderer.raymarching, psh.SHEncoder = raymarching_module(), SHEncoder
        network.sh_encode_cat = lambda enc, d, tail: torch.cat([enc(d), tail], dim=-1)
        return self

    def __exit__(self, *exc):
        renderer, psh, network = self._mods
        renderer.raymarching, psh.SHEncoder, network.sh_encode_cat = self._saved
        return False
