"""Synthetic, seeded inputs for the hot path (no datasets or checkpoints exist offline):
the analytic "brick" occupancy S0, look-at cameras and the pinhole ray generator.

Definitions follow SURVEY.md section 8(d) / Appendix B.  `get_rays` restates the full-image branch
of the reference's nerf/utils.py:52-151 (pixel centres +0.5, normalised directions, c2w rotation).
"""
import math

import numpy as np
import torch

LEGO_CAMERA_ANGLE_X = 0.6911112070083618  # NeRF-synthetic lego transforms_*.json (dataset fact)


def morton3d_np(x, y, z, bits=10):
    """Bit-by-bit interleave (independent of the magic-number form used by the kernels)."""
    x, y, z = (np.asarray(v, dtype=np.uint32) for v in (x, y, z))
    out = np.zeros_like(x)
    for b in range(bits):
        out |= ((x >> b) & 1) << (3 * b)
        out |= ((y >> b) & 1) << (3 * b + 1)
        out |= ((z >> b) & 1) << (3 * b + 2)
    return out


def brick_density_grid(H=128, seed=0, extent=0.65, bound=2):
    """Scene S0: density_grid float32 [cascade, H^3] in morton order (1 = occupied).

    cascade-0 cell (i,j,k) is occupied iff its centre p = (idx+0.5)/H*2-1 satisfies max|p| < extent
    and brick_hash(i>>3, j>>3, k>>3) % 4 != 0; a cascade-c cell is occupied iff it contains the
    centre of an occupied cascade-0 cell."""
    cascade = 1 + math.ceil(math.log2(bound))
    idx = np.arange(H, dtype=np.uint32)
    i, j, k = np.meshgrid(idx, idx, idx, indexing="ij")
    c = (idx.astype(np.float64) + 0.5) / H * 2 - 1
    px, py, pz = np.meshgrid(c, c, c, indexing="ij")
    inside = np.maximum(np.maximum(np.abs(px), np.abs(py)), np.abs(pz)) < extent
    a, b, cc = i >> 3, j >> 3, k >> 3
    h = (a * np.uint32(73856093)) ^ (b * np.uint32(19349663)) ^ (cc * np.uint32(83492791)) ^ np.uint32((seed * 2654435761) & 0xFFFFFFFF)
    occ = inside & ((h % 4) != 0)
    grid = np.zeros((cascade, H ** 3), dtype=np.float32)
    grid[0, morton3d_np(i[occ], j[occ], k[occ])] = 1.0
    for cas in range(1, cascade):
        s = float(min(2 ** cas, bound))
        qi = np.floor((px[occ] / s + 1) / 2 * H).astype(np.uint32)
        qj = np.floor((py[occ] / s + 1) / 2 * H).astype(np.uint32)
        qk = np.floor((pz[occ] / s + 1) / 2 * H).astype(np.uint32)
        grid[cas, morton3d_np(qi, qj, qk)] = 1.0
    return grid


def sparse_density_grid(H=128, seed=0, extent=0.8, bound=2, fill=0.06):
    """Scene S1: like S0 but mostly air -- 4^3-cell bricks inside max|p| < extent are occupied with probability `fill`
    (thin, scattered structure: the occupied box is ~94 % empty), coarser cascades derived as in S0."""
    cascade = 1 + math.ceil(math.log2(bound))
    idx = np.arange(H, dtype=np.uint32)
    i, j, k = np.meshgrid(idx, idx, idx, indexing="ij")
    c = (idx.astype(np.float64) + 0.5) / H * 2 - 1
    px, py, pz = np.meshgrid(c, c, c, indexing="ij")
    inside = np.maximum(np.maximum(np.abs(px), np.abs(py)), np.abs(pz)) < extent
    a, b, cc = i >> 2, j >> 2, k >> 2
    h = (a * np.uint32(73856093)) ^ (b * np.uint32(19349663)) ^ (cc * np.uint32(83492791)) ^ np.uint32((seed * 2654435761 + 12345) & 0xFFFFFFFF)
    h = (h ^ (h >> np.uint32(13))) * np.uint32(0x5bd1e995)
    occ = inside & (((h >> np.uint32(8)) % np.uint32(10000)) < np.uint32(int(fill * 10000)))
    grid = np.zeros((cascade, H ** 3), dtype=np.float32)
    grid[0, morton3d_np(i[occ], j[occ], k[occ])] = 1.0
    for cas in range(1, cascade):
        s = float(min(2 ** cas, bound))
        qi = np.floor((px[occ] / s + 1) / 2 * H).astype(np.uint32)
        qj = np.floor((py[occ] / s + 1) / 2 * H).astype(np.uint32)
        qk = np.floor((pz[occ] / s + 1) / 2 * H).astype(np.uint32)
        grid[cas, morton3d_np(qi, qj, qk)] = 1.0
    return grid


def slab_density_grid(H=128, half_thickness=0.8, bound=2):
    """Frozen slab occupancy |z| < half_thickness used for the forward-facing training config (3)."""
    cascade = 1 + math.ceil(math.log2(bound))
    idx = np.arange(H, dtype=np.uint32)
    i, j, k = np.meshgrid(idx, idx, idx, indexing="ij")
    grid = np.zeros((cascade, H ** 3), dtype=np.float32)
    for cas in range(cascade):
        s = float(min(2 ** cas, bound))
        cz = ((idx.astype(np.float64) + 0.5) / H * 2 - 1) * s
        occ = np.broadcast_to((np.abs(cz) < half_thickness)[None, None, :], i.shape)
        grid[cas, morton3d_np(i[occ], j[occ], k[occ])] = 1.0
    return grid


def packbits_np(grid, thresh):
    """NumPy statement of packbits (bit i of byte n <-> cell 8n+i, strict '>')."""
    return np.packbits((np.asarray(grid, np.float32).reshape(-1) > np.float32(thresh)), bitorder="little")


def lookat_pose(radius=4.031128874 * 0.8, elevation_deg=30.0, azimuth_deg=45.0):
    """4x4 camera-to-world looking at the origin; world up +y; columns (right, down, forward)."""
    el, az = math.radians(elevation_deg), math.radians(azimuth_deg)
    eye = np.array([radius * math.cos(el) * math.sin(az), radius * math.sin(el), radius * math.cos(el) * math.cos(az)])
    fwd = -eye / np.linalg.norm(eye)
    up = np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, down, fwd, eye
    return pose


def intrinsics_from_fov(H, W, camera_angle_x=LEGO_CAMERA_ANGLE_X):
    fl = 0.5 * W / math.tan(0.5 * camera_angle_x)
    return np.array([fl, fl, W / 2, H / 2], dtype=np.float32)


@torch.no_grad()
def get_rays(poses, intrinsics, H, W):
    """Full-image rays: poses [B,4,4] c2w, intrinsics (fx,fy,cx,cy) -> rays_o, rays_d [B, H*W, 3] (on poses.device).

    Same formulas as the reference's get_rays (pixel centres +0.5, normalise, rotate), but evaluated in float64 with
    element-wise NumPy arithmetic and rounded once to float32: torch's CPU kernels (vectorised norm, BLAS matmul) are not
    bit-reproducible across host CPUs, and a last-bit change of a direction changes march counts at full frame size."""
    device = poses.device
    P = poses.detach().cpu().numpy().astype(np.float64)
    B = P.shape[0]
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    jj, ii = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    xs = ((ii.reshape(-1) + 0.5) - cx) / fx
    ys = ((jj.reshape(-1) + 0.5) - cy) / fy
    inv = 1.0 / np.sqrt(xs * xs + ys * ys + 1.0)
    xs, ys, zs = xs * inv, ys * inv, inv
    rays_d = np.empty((B, H * W, 3), dtype=np.float32)
    rays_o = np.empty((B, H * W, 3), dtype=np.float32)
    for b in range(B):
        R = P[b, :3, :3]
        for k in range(3):
            rays_d[b, :, k] = (xs * R[k, 0] + ys * R[k, 1] + zs * R[k, 2]).astype(np.float32)
            rays_o[b, :, k] = np.float32(P[b, k, 3])
    return torch.from_numpy(rays_o).to(device), torch.from_numpy(rays_d).to(device)


def seed_field_(model, seed=0, table_range=0.5):
    """Deterministic weights: default nn.Linear init under `seed`, hash tables ~ U(-r, r)
    (the reference's U(-1e-4,1e-4) init gives sigma == 1 everywhere, useless for a benchmark)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(model.named_parameters()):
            if name.endswith("embeddings"):
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * table_range)
            elif name == "basis_color":
                p.copy_(torch.rand(p.shape, generator=g) * 0.8 + 0.1)
            elif p.dim() == 2:
                bound = 1.0 / math.sqrt(p.shape[1])
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)
            else:
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.1)
    return model


def psnr(a, b):
    """-10 log10(mean((a-b)^2)) (nerf/utils.py:242)."""
    mse = float(torch.mean((a.double() - b.double()) ** 2))
    return float("inf") if mse == 0 else -10.0 * math.log10(mse)
