"""Parity at BASELINE.json's full size (800x800 = 640 000 rays, scene S0) through size-independent properties and
checksums: the oracle needs ~40 s of CPU for this march, so its per-ray counts are pinned here by constants computed
once in the build container (oracle.march_rays_train on the same rays; see the comment next to each constant)."""
import zlib

import numpy as np
import pytest
import torch

from palettenerf_amd import network, raymarching, scene

pytestmark = pytest.mark.gpu

# oracle.march_rays_train(800x800 S0 rays, bound 2, dt_gamma 0, max_steps 1024): counter, #rays with samples, max count, crc32(counts)
ORACLE_COUNTER = 63001854
ORACLE_HITTING_RAYS = 354841
ORACLE_MAX_COUNT = 509
ORACLE_COUNTS_CRC32 = 2971143974
RAYS_D_CRC32, RAYS_O_CRC32 = 540470127, 1262356950  # the deterministic (float64 -> float32) ray generator gives these bits on any host


@pytest.fixture(scope="module")
def frame800(cuda):
    grid = torch.from_numpy(scene.brick_density_grid()).to(cuda)
    bitfield = raymarching.packbits(grid, 0.5)
    pose = torch.from_numpy(scene.lookat_pose())[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(800, 800), 800, 800)
    assert zlib.crc32(rd[0].numpy().tobytes()) == RAYS_D_CRC32 and zlib.crc32(ro[0].numpy().tobytes()) == RAYS_O_CRC32
    return grid, bitfield, ro[0].to(cuda).contiguous(), rd[0].to(cuda).contiguous()


def test_full_frame_training_march_counts_and_structure(cuda, frame800):
    grid, bitfield, ro, rd = frame800
    N = ro.shape[0]
    aabb = torch.tensor([-2, -2, -2, 2, 2, 2], dtype=torch.float32, device=cuda)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=cuda)
    M = 64 * 1024 * 1024  # mean_count path: M rows (2 GB of staging), nothing is dropped since counter < M
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 2.0, bitfield, 2, 128, nears, fars, counter, M - 128, False, 128, False, 0.0, 1024)
    assert xyzs.shape[0] == M
    counts = rays[:, 2].cpu().numpy()
    assert counter.cpu().tolist() == [ORACLE_COUNTER, N]
    assert int((counts > 0).sum()) == ORACLE_HITTING_RAYS and int(counts.max()) == ORACLE_MAX_COUNT
    assert zlib.crc32(counts.astype(np.int32).tobytes()) == ORACLE_COUNTS_CRC32          # every per-ray count equals the oracle's
    # structure: row n is ray n, offsets are the exclusive prefix sum of the counts
    assert torch.equal(rays[:, 0], torch.arange(N, dtype=torch.int32, device=cuda))
    assert torch.equal(rays[:, 1].long(), torch.cumsum(rays[:, 2].long(), 0) - rays[:, 2].long())
    m = ORACLE_COUNTER
    assert bool((deltas[:m, 0] > 0).all()) and bool((deltas[m:m + 4096] == 0).all())
    # every emitted sample lies in an occupied cell of the cascade the kernel chose for it
    p = xyzs[:m]
    mx = p.abs().amax(dim=1)
    level = (torch.frexp(mx).exponent.clamp(min=0, max=1)).long()  # dt is constant and small: the level comes from the position
    mb = torch.minimum(torch.exp2(level.float()), torch.tensor(2.0, device=cuda))[:, None]
    cell = (0.5 * (p / mb + 1) * 128).clamp(0, 127).int()
    index = level * 128 ** 3 + raymarching.morton3D(cell).long()
    occ = (bitfield[index // 8].int() >> (index % 8).int()) & 1
    assert bool(occ.all())
    # directions are the rays' own, repeated count times
    ray_of_row = torch.repeat_interleave(torch.arange(N, device=cuda), rays[:, 2].long())
    assert torch.equal(dirs[:m], rd[ray_of_row])


def test_full_frame_native_loop_equals_reference_style_loop(cuda, frame800):
    grid, bitfield, ro, rd = frame800
    m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=100.0, min_near=0.2)
    scene.seed_field_(m, 0)
    m = m.to(cuda).eval()
    m.density_grid.copy_(grid)
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    m.count_rendered = True
    out = {}
    for mode in ("compat", "native"):
        m.march_mode = mode
        m.fused_field = mode == "native"
        with torch.no_grad():
            out[mode] = m.render(ro[None], rd[None], perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    a, b = out["compat"], out["native"]
    assert int(a["rendered"].item()) == int(b["rendered"].item())  # same schedule, same compaction, same samples
    assert 9_000_000 < int(b["rendered"].item()) < 10_500_000
    assert float((a["image"] - b["image"]).abs().max()) < 2e-5
    assert float((a["weights_sum"] - b["weights_sum"]).abs().max()) < 2e-5
    assert scene.psnr(a["image"], b["image"]) > 90.0
    fin = torch.isfinite(a["depth"])
    assert torch.equal(fin, torch.isfinite(b["depth"])) and float((a["depth"][fin] - b["depth"][fin]).abs().max()) < 1e-4
    # idempotence: rendering the same frame twice gives bit-identical results (no atomics on the inference path)
    with torch.no_grad():
        c = m.render(ro[None], rd[None], perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    assert torch.equal(b["image"], c["image"]) and torch.equal(b["weights_sum"], c["weights_sum"])


def test_full_size_grid_encode_linearity_and_checksum(cuda):
    """Hash-grid forward at B = 2^20: linear in the table (f(a T1 + b T2) = a f(T1) + b f(T2) up to rounding) and
    permutation-equivariant in the samples."""
    from palettenerf_amd import gridencoder
    g = torch.Generator(device="cpu").manual_seed(3)
    enc = gridencoder.GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=4096).to(cuda)
    B = 1 << 20
    x = torch.rand(B, 3, generator=g).to(cuda)
    T1 = (torch.rand(enc.embeddings.shape, generator=g) - 0.5).to(cuda)
    T2 = (torch.rand(enc.embeddings.shape, generator=g) - 0.5).to(cuda)
    f = lambda T: gridencoder.grid_encode(x, T, enc.offsets, enc.per_level_scale, 16, False, 0, False)
    y1, y2, y12 = f(T1), f(T2), f(0.5 * T1 - 2.0 * T2)
    assert float((y12 - (0.5 * y1 - 2.0 * y2)).abs().max()) < 2e-6
    perm = torch.randperm(B, generator=g).to(cuda)
    yp = gridencoder.grid_encode(x[perm].contiguous(), T1, enc.offsets, enc.per_level_scale, 16, False, 0, False)
    assert torch.equal(yp, y1[perm])


def test_garden_video_frame_native_equals_reference_style_loop(cuda):
    """BASELINE configs[4] at its full size on one GPU: one pose of the 120-pose garden path (1297 x 840 = 1.09 M rays, PaletteNeRF,
    dt_gamma = 1/128, scene S2) through the device-driven loop against the host-driven loop with torch MLPs and the reference's seven
    composites (size-independent properties: same samples, same maps); then the 8-way tile sharding of the same frame reassembles it."""
    from palettenerf_amd import dist as pdist, rays as prays, renderer
    from palettenerf_amd.fused import tile_ray_order
    H, W = scene.GARDEN_H, scene.GARDEN_W
    opt = renderer.default_opt()
    m = network.PaletteNetwork(opt, bound=2, cuda_ray=True, density_scale=100.0, min_near=0.2)
    scene.seed_field_(m, 0)
    m = m.to(cuda).eval()
    m.density_grid.copy_(torch.from_numpy(scene.garden_density_grid()).to(cuda))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    m.count_rendered = True
    pose = torch.from_numpy(scene.garden_orbit_pose(17)[None]).to(cuda)
    ro, rd = prays.rays_from_indices(pose, scene.garden_intrinsics(), H, W, None)
    kw = dict(perturb=False, dt_gamma=1.0 / 128, max_steps=1024, T_thresh=1e-4, gui_mode=False)
    out = {}
    for mode in ("compat", "native"):
        m.march_mode = mode
        m.fused_field = mode == "native"
        with torch.no_grad():
            out[mode] = m.render(ro, rd, **kw)
    a, b = out["compat"], out["native"]
    na, nb_ = int(a["rendered"].sum()), int(b["rendered"].sum())
    # same schedule and compaction; the two loops evaluate sigma with different roundings (rocBLAS fp32 GEMMs vs the fused split-fp16 field),
    # so a handful of the 31 M samples fall on the other side of the T < 1e-4 termination test
    assert na > 5_000_000 and abs(na - nb_) <= 1e-5 * na, (na, nb_)
    for k in ("image", "weights_sum", "direct_rgb", "view_dep_rgb", "basis_rgb", "unscaled_basis_rgb", "basis_acc"):
        assert float((a[k] - b[k]).abs().max()) < 1e-4, k
    assert scene.psnr(a["image"], b["image"]) > 80.0
    fin = torch.isfinite(a["depth"])
    assert torch.equal(fin, torch.isfinite(b["depth"])) and float((a["depth"][fin] - b["depth"][fin]).abs().max()) < 2e-4
    hit = float((b["weights_sum"] > 0.5).float().mean())
    assert 0.3 < hit <= 1.0   # the ground slab and the object fill a large part of every view of the path
    # the 8-GPU split of this frame on one GPU: every tile shard rendered on its own (tile-ordered alive list, as bench.py does) gives its
    # pixels of the full frame bit for bit -- per-ray results do not depend on which rays share a launch
    full = b["image"][0]
    n_total = 0
    for rank in range(8):
        idx, _ = pdist.shard_indices(H, W, rank, 8)
        idx = idx.to(cuda)
        m._fused.ray_order = tile_ray_order(idx.cpu(), W, 8).to(cuda)
        with torch.no_grad():
            r = m.render(ro[:, idx].contiguous(), rd[:, idx].contiguous(), **kw)
        assert torch.equal(r["image"][0], full[idx])
        n_total += int(r["rendered"].sum())
    m._fused.ray_order = None
    # (march-emitted samples depend on the n_step schedule, which follows each launch's own N / n_alive: samples marched for a ray after it
    # terminated inside a group are counted but never composited -- the pixels above are what must agree)
    assert abs(n_total - int(b["rendered"].sum())) < 0.01 * n_total
