"""On-device ray generation: same call surface as the reference's `get_rays` (nerf/utils.py:53-149).

The pixel-index selection is torch RNG logic on the device (as in the reference: torch.randint / torch.multinomial);
indices -> (rays_o, rays_d) is one HIP launch (`pnr_get_rays`) instead of a dozen elementwise / gather / matmul kernels.
"""
import ctypes

import torch

from ._torch_glue import call, ptr, require


def rays_from_indices(poses, intrinsics, H, W, inds=None):
    """poses [B,4,4] fp32 on the GPU, inds [B,N] int64 (or None = every pixel, row-major) -> rays_o, rays_d [B,N,3]."""
    poses = require(poses.contiguous(), torch.float32, "poses")
    B = poses.shape[0]
    if inds is not None:
        inds = require(inds.expand(B, inds.shape[-1]).contiguous(), torch.int64, "inds")
    N = H * W if inds is None else inds.shape[1]
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    rays_o = torch.empty(B, N, 3, dtype=torch.float32, device=poses.device)
    rays_d = torch.empty_like(rays_o)
    call("pnr_get_rays", ptr(poses), ctypes.c_uint32(B), ctypes.c_float(fx), ctypes.c_float(fy), ctypes.c_float(cx), ctypes.c_float(cy),
         ctypes.c_uint32(H), ctypes.c_uint32(W), ptr(inds) if inds is not None else None, ctypes.c_uint32(N), ptr(rays_o), ptr(rays_d))
    return rays_o, rays_d


def _patch_indices(H, W, count, patch, device):
    """Top-left corners drawn uniformly, each expanded to a patch x patch block (utils.py:79-96)."""
    n_patches = count // (patch * patch)
    top = torch.randint(0, H - patch, size=[n_patches], device=device)
    left = torch.randint(0, W - patch, size=[n_patches], device=device)
    dy, dx = torch.meshgrid(torch.arange(patch, device=device), torch.arange(patch, device=device), indexing="ij")
    yy = top[:, None] + dy.reshape(1, -1)
    xx = left[:, None] + dx.reshape(1, -1)
    return (yy * W + xx).reshape(-1)


def _pair_indices(H, W, count, reach, device):
    """count/2 random pixels followed by one jittered partner each (utils.py:97-111)."""
    assert count % 2 == 0
    half = count // 2
    y0 = torch.randint(0, H, size=[half], device=device)
    x0 = torch.randint(0, W, size=[half], device=device)
    jy = torch.randint(-reach, reach, size=[half], device=device)
    jx = torch.randint(-reach, reach, size=[half], device=device)
    y1 = (y0 + jy).clamp(0, H - 1)
    x1 = (x0 + jx).clamp(0, W - 1)
    return torch.cat([y0 * W + x0, y1 * W + x1], dim=0)


def get_rays(poses, intrinsics, H, W, N=-1, error_map=None, patch_size=1, random_size=0):
    """Drop-in for nerf/utils.py:get_rays: returns {'rays_o', 'rays_d', 'inds'[, 'inds_coarse']} with [B,N,3] / [B,N] tensors.
    Random draws follow the reference's call order, so the same torch seed selects the same pixels."""
    device = poses.device
    B = poses.shape[0]
    out = {}
    if N > 0:
        N = min(N, H * W)
        if patch_size > 1:
            inds = _patch_indices(H, W, N, patch_size, device).expand(B, -1)
        elif random_size > 0:
            inds = _pair_indices(H, W, N, random_size, device).expand(B, -1)
        elif error_map is None:
            inds = torch.randint(0, H * W, size=[N], device=device).expand(B, N)
        else:
            coarse = torch.multinomial(error_map.to(device), N, replacement=False)  # [B, N] cells of the 128 x 128 error grid
            cy, cx = coarse // 128, coarse % 128
            sy, sx = H / 128, W / 128
            yy = (cy * sy + torch.rand(B, N, device=device) * sy).long().clamp(max=H - 1)
            xx = (cx * sx + torch.rand(B, N, device=device) * sx).long().clamp(max=W - 1)
            inds = yy * W + xx
            out["inds_coarse"] = coarse
        rays_o, rays_d = rays_from_indices(poses, intrinsics, H, W, inds)
    else:
        inds = torch.arange(H * W, device=device).expand(B, H * W)
        rays_o, rays_d = rays_from_indices(poses, intrinsics, H, W, None)
    out["inds"] = inds
    out["rays_o"] = rays_o
    out["rays_d"] = rays_d
    return out


def image_to_uint8(image, linear_to_srgb=False):
    """What the reference does on the host before writing a frame (nerf/utils.py:719-723, 1010-1017): `(pred * 255).astype(np.uint8)`, after
    `linear_to_srgb` for scenes in linear colour -- here on the device (pnr_image_to_uint8), so a frame crosses PCIe as bytes."""
    src = require(image.contiguous(), torch.float32, "image")
    out = torch.empty(src.shape, dtype=torch.uint8, device=src.device)
    call("pnr_image_to_uint8", ptr(src), ctypes.c_uint64(src.numel()), ctypes.c_int(int(bool(linear_to_srgb))), ptr(out))
    return out
