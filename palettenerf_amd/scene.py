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


# ---------------------------------------------------------------------------------------------- scene S2 ("garden-like")
# Mip-NeRF-360 garden, images_4: 1297 x 840 pixels, focal ~ 961 px (dataset facts, not in the reference tree).
GARDEN_W, GARDEN_H, GARDEN_FL = 1297, 840, 961.0
GARDEN_POSES = 120        # scripts/llff2nerf.py:106: gen_renderposes(poses, bounds, 120)


def garden_density_grid(H=128, seed=0, bound=2):
    """Scene S2: an unbounded-360-style layout inside bound = 2 (the mip360 configs: bound 2, scale 0.12-0.16, so the cameras orbit at
    ~0.5 from the origin): a bricked centre object (max|p| < 0.3, 4^3-cell bricks, 3 of 4 present), a thin ground slab through the whole
    box, scattered "hedge" bricks around the object and a sparse far shell that only the coarse cascade resolves.  Morton order, 1 = occupied;
    cascade 1 = every cell that contains the centre of an occupied cascade-0 cell, plus the ground and the far shell outside |p| < 1."""
    cascade = 1 + math.ceil(math.log2(bound))
    assert cascade == 2, "S2 is defined for bound = 2"
    idx = np.arange(H, dtype=np.uint32)
    i, j, k = np.meshgrid(idx, idx, idx, indexing="ij")
    c = (idx.astype(np.float64) + 0.5) / H * 2 - 1
    px, py, pz = np.meshgrid(c, c, c, indexing="ij")

    def bhash(a, b, cc, salt):
        h = (a * np.uint32(73856093)) ^ (b * np.uint32(19349663)) ^ (cc * np.uint32(83492791)) ^ np.uint32((seed * 2654435761 + salt) & 0xFFFFFFFF)
        return (h ^ (h >> np.uint32(13))) * np.uint32(0x5bd1e995)

    m = np.maximum(np.maximum(np.abs(px), np.abs(py)), np.abs(pz))
    obj = (m < 0.3) & ((bhash(i >> 2, j >> 2, k >> 2, 1) >> np.uint32(8)) % np.uint32(4) != 0)
    ground = (py > -0.42) & (py < -0.34)
    hedge = (np.maximum(np.abs(px), np.abs(pz)) > 0.55) & (py >= -0.34) & (py < 0.25) & ((bhash(i >> 3, j >> 3, k >> 3, 2) >> np.uint32(8)) % np.uint32(12) == 0)
    occ0 = obj | ground | hedge
    grid = np.zeros((cascade, H ** 3), dtype=np.float32)
    grid[0, morton3d_np(i[occ0], j[occ0], k[occ0])] = 1.0
    s = 2.0
    qi = np.floor((px[occ0] / s + 1) / 2 * H).astype(np.uint32)
    qj = np.floor((py[occ0] / s + 1) / 2 * H).astype(np.uint32)
    qk = np.floor((pz[occ0] / s + 1) / 2 * H).astype(np.uint32)
    grid[1, morton3d_np(qi, qj, qk)] = 1.0
    qx, qy, qz = px * s, py * s, pz * s          # cascade-1 cell centres
    outside = np.maximum(np.maximum(np.abs(qx), np.abs(qy)), np.abs(qz)) >= 1.0
    far_ground = outside & (qy > -0.44) & (qy < -0.32)
    far_shell = outside & (qy >= -0.32) & (qy < 0.9) & ((bhash(i >> 3, j >> 3, k >> 3, 3) >> np.uint32(8)) % np.uint32(8) == 0)
    occ1 = far_ground | far_shell
    grid[1, morton3d_np(i[occ1], j[occ1], k[occ1])] = 1.0
    return grid


def lookat_pose_from(eye, target=(0.0, 0.0, 0.0)):
    """4x4 camera-to-world at `eye` looking at `target`; world up +y; columns (right, down, forward) as lookat_pose."""
    eye = np.asarray(eye, np.float64)
    fwd = np.asarray(target, np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.array([0.0, 1.0, 0.0]))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, down, fwd, eye
    return pose


def garden_orbit_pose(i, n=GARDEN_POSES, radius=0.52, height=0.22):
    """Pose i of the n-pose ellipse path of the video render (the role of nerf_360_v2.gen_renderposes, scripts/llff2nerf.py:104-106):
    an ellipse of semi-axes radius x 0.8 radius around the object at `height`, looking slightly below the origin."""
    a = 2.0 * math.pi * (i % n) / n
    return lookat_pose_from((radius * math.sin(a), height, 0.8 * radius * math.cos(a)), (0.0, -0.1, 0.0))


def garden_intrinsics(H=GARDEN_H, W=GARDEN_W):
    fl = GARDEN_FL * W / GARDEN_W
    return np.array([fl, fl, W / 2, H / 2], dtype=np.float32)


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


def stylizer_state(nb, seed):
    """Seeded, clearly non-trivial Stylizer parameters {dI [nb], dP [1,nb,3], ddelta [nb,3,3]} (palette/renderer.py:150-165 initialises them to
    zero / identity; the GUI's optimiser moves them).  Used by the fixtures (tests/golden/gen_golden.py) and the tests that replay them."""
    g = torch.Generator().manual_seed(seed)
    return {"dI": (torch.rand(nb, generator=g) - 0.5) * 0.6, "dP": (torch.rand(1, nb, 3, generator=g) - 0.5) * 0.4,
            "ddelta": torch.eye(3)[None].repeat(nb, 1, 1) + (torch.rand(nb, 3, 3, generator=g) - 0.5) * 0.5}


def psnr(a, b):
    """-10 log10(mean((a-b)^2)) (nerf/utils.py:242)."""
    mse = float(torch.mean((a.double() - b.double()) ** 2))
    return float("inf") if mse == 0 else -10.0 * math.log10(mse)
