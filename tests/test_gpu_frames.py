"""End-to-end parity on the GPU: the HIP operators under this repo's renderer/network mirror must
reproduce the golden frames that the REFERENCE's Python callers produced with the CPU oracle
injected (tests/golden/gen_golden.py).  fp32 colour tolerance from the north star: 1e-4 absolute."""
import os

import numpy as np
import pytest
import torch

from palettenerf_amd import network, raymarching, renderer, scene

pytestmark = pytest.mark.gpu

COLOUR_TOL = 1e-4
# Round 2 checked edited images at 3e-4, depth at 2e-4 and depth_origin at 5e-4 "because the hue wrap amplifies rounding".  Measured in round 3
# (profiles/edit_tolerance.py, profiles/r03_edit_tolerance.txt): every mode is within 9.5e-7 of the goldens on the images, edited or not,
# 5.3e-7 on depth and 2.9e-6 on depth_origin (values up to 2.4); no pixel is above 2e-5.  The looser bounds were never needed.
EDIT_TOL = 1e-4
DEPTH_TOL = 1e-4
# Gradients against the reference-driven goldens, relative to the largest entry of each gradient.  Rounds 1-3 allowed 2e-3; measured in round 4
# (profiles/grad_tolerance.py, profiles/r04_grad_tolerance.txt): 4e-7 ... 3.8e-5 (the worst: the table gradient of the translucent case b, whose
# largest entry is 2.4e-8 -- fp32 sums of ~1e4 addends in a different order).  The bound is 4 x the worst measurement.
GRAD_REL_TOL = 1.6e-4


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def frame_rays(g, cuda):
    H, W = int(g["H"]), int(g["W"])
    pose = torch.from_numpy(scene.lookat_pose())[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(H, W), H, W)
    return ro.to(cuda), rd.to(cuda)


def put_scene(model, cuda):
    grid = torch.from_numpy(scene.brick_density_grid()).to(cuda)
    model.density_grid.copy_(grid)
    raymarching.packbits(model.density_grid, 0.5, model.density_bitfield)  # the HIP packbits produces the bitfield


def close(got, want, tol=COLOUR_TOL, what=""):
    got = got.detach().cpu().numpy()
    m = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), m), what
    err = np.abs(got[m] - want[m]).max() if m.any() else 0.0
    assert err <= tol, f"{what}: max abs err {err}"


@pytest.mark.parametrize("case", ["a", "b"])
@pytest.mark.parametrize("mode", ["compat", "device", "fused", "native", "native_fp32"])
def test_nerf_inference_frame(cuda, golden_dir, case, mode):
    g = load(golden_dir, f"frame_nerf_{case}")
    m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=float(g["density_scale"]), min_near=0.2)
    scene.seed_field_(m, int(g["seed"]))
    m = m.to(cuda).eval()
    put_scene(m, cuda)
    m.march_mode = {"fused": "device", "native_fp32": "native"}.get(mode, mode)
    m.fused_field = mode in ("fused", "native", "native_fp32")
    m.count_rendered = True
    if mode == "native_fp32":
        from palettenerf_amd.fused import NeRFFieldFused
        m._fused = NeRFFieldFused(m)
        m._fused.precision = 0
    ro, rd = frame_rays(g, cuda)
    with torch.no_grad():
        r = m.render(ro, rd, staged=True, dt_gamma=float(g["dt_gamma"]), perturb=False, max_steps=1024, T_thresh=1e-4)
    close(r["image"], g["image"], what="image")
    close(r["weights_sum"], g["weights_sum"], what="weights_sum")
    close(r["depth"], g["depth"], tol=DEPTH_TOL, what="depth")
    assert scene.psnr(r["image"].cpu(), torch.from_numpy(g["image"])) > 80.0
    _RENDERED.setdefault(case, {})[mode] = (int(r["rendered"].item()), int(r["n_samples"]))
    if len(_RENDERED[case]) == 5:  # every execution mode marched exactly the same samples (schedule and compaction identical)
        counts = {k: v[0] for k, v in _RENDERED[case].items()}
        assert len(set(counts.values())) == 1, counts


_RENDERED = {}


@pytest.mark.parametrize("case", ["a", "b"])
def test_dropin_fuse_field_under_the_reference_style_loop(cuda, golden_dir, case):
    """dropin.fuse_field(model): the per-op loop (march_rays / composite_rays / boolean-mask compaction, what an unchanged run_cuda issues) with
    `self(xyzs, dirs)` served by the fused MFMA field -- against the reference-driven golden frame, and the same samples as every other mode."""
    from palettenerf_amd import dropin
    g = load(golden_dir, f"frame_nerf_{case}")
    m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=float(g["density_scale"]), min_near=0.2)
    scene.seed_field_(m, int(g["seed"]))
    m = m.to(cuda).eval()
    put_scene(m, cuda)
    m.count_rendered = True
    m.march_mode = "compat"
    ro, rd = frame_rays(g, cuda)
    with torch.no_grad():
        plain = m.render(ro, rd, staged=True, dt_gamma=float(g["dt_gamma"]), perturb=False, max_steps=1024, T_thresh=1e-4)
    assert dropin.fuse_field(m) is m and m.fused_field is False            # only forward() changed; the renderer's own switches are untouched
    with torch.no_grad():
        r = m.render(ro, rd, staged=True, dt_gamma=float(g["dt_gamma"]), perturb=False, max_steps=1024, T_thresh=1e-4)
    close(r["image"], g["image"], what="image")
    close(r["weights_sum"], g["weights_sum"], what="weights_sum")
    close(r["depth"], g["depth"], tol=DEPTH_TOL, what="depth")
    assert int(r["rendered"].item()) == int(plain["rendered"].item())      # same march, same termination
    x = torch.rand(64, 3, device=cuda, requires_grad=True)                  # under autograd the model's own forward runs (and differentiates)
    s_, c_ = m(x * 2 - 1, torch.nn.functional.normalize(torch.randn(64, 3, device=cuda), dim=-1))
    assert s_.requires_grad and c_.requires_grad


@pytest.mark.parametrize("case", ["a", "b"])
def test_nerf_training_step(cuda, golden_dir, case):
    g = load(golden_dir, f"train_nerf_{case}")
    m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=float(g["density_scale"]), min_near=0.2)
    scene.seed_field_(m, int(g["seed"]))
    m = m.to(cuda).train()
    put_scene(m, cuda)
    ro, rd = frame_rays(g, cuda)
    r = m.run_cuda(ro, rd, dt_gamma=float(g["dt_gamma"]), perturb=False, force_all_rays=True, max_steps=1024, T_thresh=1e-4)
    assert m.step_counter[0].cpu().numpy().tolist() == g["counter"].tolist()  # sample / ray counts: bit-exact
    close(r["image"], g["image"], what="image")
    close(r["weights_sum"], g["weights_sum"], what="weights_sum")
    close(r["depth"], g["depth"], tol=DEPTH_TOL, what="depth")
    loss = (r["image"] ** 2).mean() + 0.1 * r["weights_sum"].mean()
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    scale = lambda a: max(1e-6, float(np.abs(a).max()))
    for got, key in ((m.color_net[0].weight.grad, "grad_color0"), (m.sigma_net[1].weight.grad, "grad_sigma1")):
        close(got, g[key], tol=GRAD_REL_TOL * scale(g[key]), what=key)
    rows = torch.from_numpy(g["grad_emb_rows"]).to(cuda)
    close(m.encoder.embeddings.grad[rows], g["grad_emb_vals"], tol=GRAD_REL_TOL * scale(g["grad_emb_vals"]), what="grad_emb")
    assert abs(float(m.encoder.embeddings.grad.abs().sum()) / float(g["grad_emb_abs_sum"]) - 1) < 1e-3


@pytest.mark.parametrize("case", ["a", "b"])
def test_palette_inference_frame_all_maps_and_edit(cuda, golden_dir, case):
    g = load(golden_dir, f"frame_palette_{case}")
    opt = renderer.default_opt(pred_clip=bool(g["pred_clip"]))
    m = network.PaletteNetwork(opt, bound=2, cuda_ray=True, density_scale=float(g["density_scale"]), min_near=0.2)
    scene.seed_field_(m, int(g["seed"]))
    m = m.to(cuda).eval()
    put_scene(m, cuda)
    ro, rd = frame_rays(g, cuda)
    for mode in ("compat", "device", "fused", "native"):
        m.march_mode = "device" if mode == "fused" else mode
        m.fused_field = mode in ("fused", "native")
        with torch.no_grad():
            r = m.render(ro, rd, dt_gamma=float(g["dt_gamma"]), perturb=False, max_steps=1024, T_thresh=1e-4, gui_mode=False)
        for k in ("image", "weights_sum", "clip_feat", "direct_rgb", "view_dep_rgb", "basis_rgb", "unscaled_basis_rgb", "basis_acc"):
            close(r[k], g[k], what=f"{mode}:{k}")
        close(r["depth"], g["depth"], tol=DEPTH_TOL, what="depth")
        close(r["depth_origin"], g["depth_origin"], tol=DEPTH_TOL, what="depth_origin")
    # regional edit: RGB->HSV->RGB inside the fused field epilogue of the device-driven loop (the last mode set above)
    m.edit = renderer.RegionEdit(opt)
    m.edit.update_cent(mean_xyz=torch.tensor([0.1, 0.0, -0.2], device=cuda))
    m.edit.update_std(std_xyz=0.5)
    m.edit.update_delta_hsv(m.basis_color.data.clamp(0, 1), (m.basis_color.data * 0.6 + 0.2).flip(0).clamp(0, 1))
    with torch.no_grad():
        r2 = m.render(ro, rd, dt_gamma=float(g["dt_gamma"]), perturb=False, max_steps=1024, T_thresh=1e-4, gui_mode=True)
    close(r2["image"], g["edit_image"], tol=EDIT_TOL, what="edit_image")
    assert "basis_rgb" not in r2


@pytest.mark.parametrize("case", ["a", "b"])
def test_dropin_fuse_field_palette_under_the_reference_style_loop(cuda, golden_dir, case):
    """dropin.fuse_field on a PaletteNetwork: the per-op loop with the renderer's own colour-basis arithmetic and its seven flex composites, only
    `self(xyzs, dirs)` served by the fused kernel's network-heads row -- all maps against the reference-driven golden frame (case b: clip head)."""
    from palettenerf_amd import dropin
    g = load(golden_dir, f"frame_palette_{case}")
    opt = renderer.default_opt(pred_clip=bool(g["pred_clip"]))
    m = network.PaletteNetwork(opt, bound=2, cuda_ray=True, density_scale=float(g["density_scale"]), min_near=0.2)
    scene.seed_field_(m, int(g["seed"]))
    m = m.to(cuda).eval()
    put_scene(m, cuda)
    m.march_mode = "compat"
    dropin.fuse_field(m)
    assert m.fused_field is False
    ro, rd = frame_rays(g, cuda)
    with torch.no_grad():
        r = m.render(ro, rd, dt_gamma=float(g["dt_gamma"]), perturb=False, max_steps=1024, T_thresh=1e-4, gui_mode=False)
    for k in ("image", "weights_sum", "clip_feat", "direct_rgb", "view_dep_rgb", "basis_rgb", "unscaled_basis_rgb", "basis_acc"):
        close(r[k], g[k], what=k)
    close(r["depth"], g["depth"], tol=DEPTH_TOL, what="depth")


@pytest.mark.parametrize("case", ["a", "b"])
def test_palette_training_step(cuda, golden_dir, case):
    g = load(golden_dir, f"train_palette_{case}")
    opt = renderer.default_opt(pred_clip=bool(g["pred_clip"]))
    m = network.PaletteNetwork(opt, bound=2, cuda_ray=True, density_scale=float(g["density_scale"]), min_near=0.2)
    scene.seed_field_(m, int(g["seed"]))
    m = m.to(cuda).train()
    put_scene(m, cuda)
    ro, rd = frame_rays(g, cuda)
    r = m.run_cuda(ro, rd, dt_gamma=float(g["dt_gamma"]), perturb=False, force_all_rays=True, max_steps=1024, T_thresh=1e-4)
    assert m.step_counter[0].cpu().numpy().tolist() == g["counter"].tolist()
    for k in ("image", "weights_sum", "omega_sparsity", "view_dep_norm", "offsets_norm", "direct_rgb", "view_dep_rgb", "diffuse_rgb", "clip_feat", "basis_acc"):
        close(r[k], g[k], tol=COLOUR_TOL, what=k)
    loss = (r["image"] ** 2).mean() + 0.01 * r["omega_sparsity"].mean() + 0.1 * r["offsets_norm"].mean() + (r["direct_rgb"] ** 2).mean() \
        + 0.1 * (r["clip_feat"] ** 2).mean() + 0.1 * r["basis_acc"].mean()
    assert abs(float(loss) - float(g["loss"])) < 2e-5
    loss.backward()
    scale = lambda a: max(1e-6, float(np.abs(a).max()))
    for got, key in ((m.offsets_radiance_net.weight.grad, "grad_offsets_radiance"), (m.basis_color.grad, "grad_basis_color"), (m.diff_net[0].weight.grad, "grad_diff0")):
        close(got, g[key], tol=GRAD_REL_TOL * scale(g[key]), what=key)
    rows = torch.from_numpy(g["grad_emb_rows"]).to(cuda)
    close(m.encoder_palette.embeddings.grad[rows], g["grad_emb_vals"], tol=GRAD_REL_TOL * scale(g["grad_emb_vals"]), what="grad_emb_palette")
    assert (m.encoder.embeddings.grad is None) == bool(g["encoder_grad_is_none"])  # sigma is detached: geometry frozen


def test_occupancy_maintenance_produces_the_bitfield(cuda):
    """f1: update_extra_state / mark_untrained_grid on the HIP morton + packbits ops.  With an analytic density the swept
    grid, the EMA-max, the threshold and the packed bitfield can all be checked in closed form."""
    m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=2.0, density_thresh=0.5, min_near=0.2).to(cuda)

    def density(x):  # sigma = 1 inside the sphere |x| < 0.9, else 0: piecewise constant away from the surface
        return {"sigma": (x.norm(dim=-1) < 0.9).float()}

    m.density = density
    m.local_step, m.step_counter[:3, 0] = 3, torch.tensor([100, 200, 330], dtype=torch.int32, device=cuda)
    m.update_extra_state()
    assert m.iter_density == 1 and m.local_step == 0 and m.mean_count == 210
    G = 128
    full = torch.stack(torch.meshgrid(*[torch.arange(G, dtype=torch.int32, device=cuda)] * 3, indexing="ij"), -1).reshape(-1, 3)
    idx = raymarching.morton3D(full).long()
    for cas in range(2):
        bound = min(2 ** cas, 2)
        centre = (2 * full.float() / (G - 1) - 1) * (bound - bound / G)
        r = centre.norm(dim=-1)
        margin = 2 * bound / G * 1.8  # a jittered point stays within half a cell (x sqrt 3) of the centre
        inside, outside = r < 0.9 - margin, r > 0.9 + margin
        assert bool((m.density_grid[cas, idx[inside]] == 2.0).all())   # sigma * density_scale
        assert bool((m.density_grid[cas, idx[outside]] == 0.0).all())
    mean_density = float(m.density_grid.clamp(min=0).mean())
    assert abs(m.mean_density - mean_density) < 1e-6
    want = scene.packbits_np(m.density_grid.cpu().numpy(), min(mean_density, 0.5))
    assert np.array_equal(m.density_bitfield.cpu().numpy(), want)
    assert raymarching.occupancy_mip(m.density_bitfield, 2, 128, 2.0) is not None
    # EMA-max: a second sweep with a vanished field decays instead of clearing
    m.density = lambda x: {"sigma": torch.zeros(x.shape[0], device=x.device)}
    before = m.density_grid.clone()
    m.update_extra_state(decay=0.5)
    assert torch.allclose(m.density_grid, before * 0.5)
    # partial-update branch (iter_density >= 16) keeps the invariants
    m.iter_density = 16
    m.update_extra_state()
    assert m.iter_density == 17 and bool((m.density_grid >= 0).all())
    # mark_untrained_grid: one camera at +z looking down -z sees the cone in front of it only
    m2 = network.NeRFNetwork(bound=2, cuda_ray=True, min_near=0.2).to(cuda)
    pose = torch.eye(4)
    pose[:3, 3] = torch.tensor([0.0, 0.0, 3.0])
    pose[:3, 2] = torch.tensor([0.0, 0.0, -1.0])  # forward = -z
    pose[:3, 0] = torch.tensor([-1.0, 0.0, 0.0])
    n_marked = m2.mark_untrained_grid(pose[None], (100.0, 100.0, 50.0, 50.0))
    assert 0 < n_marked < 2 * G ** 3
    behind = raymarching.morton3D(torch.tensor([[64, 64, 127]], dtype=torch.int32, device=cuda)).long()  # z ~ +1 * (bound): in front of the camera? no: camera at z=3 looks to -z, so all cells have z < 3
    centre_cell = raymarching.morton3D(torch.tensor([[64, 64, 64]], dtype=torch.int32, device=cuda)).long()
    assert float(m2.density_grid[0, centre_cell]) == 0.0       # seen: stays trainable
    corner = raymarching.morton3D(torch.tensor([[0, 0, 127]], dtype=torch.int32, device=cuda)).long()
    assert float(m2.density_grid[1, corner]) == -1.0           # far off-axis next to the camera plane: never seen


@pytest.mark.parametrize("model_kind", ["nerf", "palette"])
def test_native_loop_edge_cases_match_reference_style_loop(cuda, model_kind):
    """Device-driven loop vs the host-driven mirror on awkward inputs: a single ray, rays that all miss the scene box,
    a step budget that ends the loop early, a translucent field (hundreds of iterations, n_step growing to 8)."""
    if model_kind == "nerf":
        m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=1.0, min_near=0.2)
    else:
        m = network.PaletteNetwork(renderer.default_opt(), bound=2, cuda_ray=True, density_scale=1.0, min_near=0.2)
    scene.seed_field_(m, 11)
    m = m.to(cuda).eval()
    put_scene(m, cuda)
    m.count_rendered = True
    pose = torch.from_numpy(scene.lookat_pose())[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(48, 48), 48, 48)
    ro, rd = ro.to(cuda), rd.to(cuda)

    def both(ro_, rd_, **kw):
        out = {}
        for mode in ("compat", "native"):
            m.march_mode, m.fused_field = mode, mode == "native"
            with torch.no_grad():
                out[mode] = m.render(ro_, rd_, perturb=False, T_thresh=1e-4, **kw)
        a, b = out["compat"], out["native"]
        assert int(a["rendered"].item()) == int(b["rendered"].item()), kw
        for k in ("image", "weights_sum"):
            close(b[k], a[k].cpu().numpy(), tol=1e-4, what=k)
        close(b["depth"], a["depth"].cpu().numpy(), tol=DEPTH_TOL, what="depth")   # NaN pattern (0/0 for missed rays) must agree too
        return a, b

    a, b = both(ro[:, 1000:1001].contiguous(), rd[:, 1000:1001].contiguous(), dt_gamma=0, max_steps=1024)       # N = 1
    a, b = both(ro[:, :7].contiguous(), rd[:, :7].contiguous(), dt_gamma=0, max_steps=1024)                     # N < one wave
    away = ro.clone()
    away[..., 1] += 50.0                                                                                             # every ray misses the +-2 box
    a, b = both(away, rd, dt_gamma=0, max_steps=1024)
    assert int(b["rendered"].item()) == 0 and float(b["weights_sum"].abs().max()) == 0.0
    a, b = both(ro, rd, dt_gamma=0, max_steps=16)                                                                 # step budget ends the loop
    assert int(b["rendered"].item()) > 0
    m.density_scale = 0.02                                                                                           # translucent: long marches
    a, b = both(ro, rd, dt_gamma=1.0 / 128, max_steps=1024)
    a, b = both(ro, rd, dt_gamma=0, max_steps=1024)
    assert b["iterations"] > 40


@pytest.mark.parametrize("dt_gamma,density_scale,max_steps", [(0.0, 1.0, 1024), (1.0 / 128, 0.02, 1024), (0.0, 0.05, 1024), (0.0, 1.0, 16), (1.0 / 32, 0.3, 256)])
@pytest.mark.parametrize("scene_kind", ["bricks", "sparse"])
def test_coop_march_tail_is_bit_identical(cuda, dt_gamma, density_scale, max_steps, scene_kind):
    """The wave-cooperative march tail (frame.hip: march_coop_tail -- the last <= 4 rays of a wave marched by all 64 lanes, one lattice
    point per lane) against the same frame with it switched off: every output bit, the sample count and the iteration count.  Opaque and
    translucent fields (n_step from 1 to 8), constant and growing steps, the degenerate dt_min > dt_max budget, dense and sparse scenes."""
    from palettenerf_amd import _lib
    lib = _lib.load()
    m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=density_scale, min_near=0.2)
    scene.seed_field_(m, 3)
    m = m.to(cuda).eval()
    grid = scene.brick_density_grid() if scene_kind == "bricks" else scene.sparse_density_grid()
    m.density_grid.copy_(torch.from_numpy(grid).to(cuda))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    m.march_mode, m.fused_field, m.count_rendered = "native", True, True
    pose = torch.from_numpy(scene.lookat_pose(azimuth_deg=70.0))[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(160, 200), 160, 200)
    ro, rd = ro.to(cuda), rd.to(cuda)
    out = []
    try:
        for coop in (1, 0):
            assert lib.pnr_set_option(b"coop_march", coop) == 0
            with torch.no_grad():
                r = m.render(ro, rd, perturb=False, dt_gamma=dt_gamma, max_steps=max_steps, T_thresh=1e-4)
            out.append({k: r[k].clone() for k in ("image", "depth", "weights_sum", "rendered")})
    finally:
        lib.pnr_set_option(b"coop_march", 1)
    a, b = out
    assert int(a["rendered"]) == int(b["rendered"]) > 1000
    for k in ("image", "depth", "weights_sum"):
        assert torch.equal(torch.nan_to_num(a[k], nan=-7.0), torch.nan_to_num(b[k], nan=-7.0)), k


@pytest.mark.parametrize("layout", ["single", "pair", "triple", "half1", "half2"])
@pytest.mark.parametrize("dt_gamma,density_scale,scene_kind", [(0.0, 1.0, "bricks"), (1.0 / 128, 0.02, "sparse"), (0.0, 0.05, "sparse")])
def test_hosted_march_tail_is_bit_identical(cuda, layout, dt_gamma, density_scale, scene_kind):
    """The hosted march tail (frame.hip: k_frame_march<.., 2> queues the rays it has not finished within its probe budget, the first
    workgroups of the lookup launch march them and look their rows up) against the same frame without it: every output bit, the sample
    count and the iteration count -- for every table layout of the lookup kernels (one fp32 table, the interleaved pair, the triple with a
    clip head, one and two fp16 tables), budgets from 1 probe (most of a frame goes through the queue) to 5, the first launch of a frame
    budgeted too, opaque and translucent fields (n_step from 1 to 8), constant and growing steps, dense and sparse scenes."""
    from palettenerf_amd import _lib
    from palettenerf_amd.fused import NeRFFieldFused, PaletteFieldFused
    lib = _lib.load()
    if layout in ("single", "half1"):
        m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=density_scale, min_near=0.2)
    else:
        m = network.PaletteNetwork(renderer.default_opt(pred_clip=layout == "triple"), bound=2, cuda_ray=True, density_scale=density_scale, min_near=0.2)
    scene.seed_field_(m, 3)
    m = m.to(cuda).eval()
    grid = scene.brick_density_grid() if scene_kind == "bricks" else scene.sparse_density_grid()
    m.density_grid.copy_(torch.from_numpy(grid).to(cuda))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    m.march_mode, m.fused_field, m.count_rendered = "native", True, True
    m._fused = (NeRFFieldFused if layout in ("single", "half1") else PaletteFieldFused)(m)
    m._fused.table_half = layout.startswith("half")
    pose = torch.from_numpy(scene.lookat_pose(azimuth_deg=70.0))[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(160, 200), 160, 200)
    ro, rd = ro.to(cuda), rd.to(cuda)
    keys = ["image", "depth", "weights_sum", "rendered", "iterations", "view_dep_rgb", "diffuse_rgb", "direct_rgb", "basis_acc", "clip_feat", "omega_sparsity"]
    out = []
    try:
        for hosted, budget, budget0 in ((0, 2, 0), (1, 2, 0), (1, 1, 0), (1, 5, 3), (1, 1, 1)):
            assert lib.pnr_set_option(b"hosted_tail", hosted) == 0
            assert lib.pnr_set_option(b"march_budget", budget) == 0 and lib.pnr_set_option(b"march_budget0", budget0) == 0
            with torch.no_grad():
                r = m.render(ro, rd, perturb=False, dt_gamma=dt_gamma, max_steps=1024, T_thresh=1e-4)
            out.append({k: torch.as_tensor(r[k]).clone() for k in keys if k in r})
    finally:
        lib.pnr_set_option(b"hosted_tail", 1); lib.pnr_set_option(b"march_budget", 2); lib.pnr_set_option(b"march_budget0", 0)
    a = out[0]
    assert int(a["rendered"]) > 1000
    for b in out[1:]:
        assert int(a["rendered"]) == int(b["rendered"])
        for k in a:
            assert torch.equal(torch.nan_to_num(a[k], nan=-7.0), torch.nan_to_num(b[k], nan=-7.0)), k


@pytest.mark.parametrize("model_kind", ["nerf", "palette"])
def test_native_loop_ray_order_leaves_every_output_bit_identical(cuda, model_kind):
    """pnr_*_frame_args::ray_order (tile order, a random permutation) changes the processing order only: image, depth, weights_sum,
    every palette map and the sample count must be bit-identical to the unordered frame."""
    from palettenerf_amd.fused import NeRFFieldFused, PaletteFieldFused, tile_ray_order
    if model_kind == "nerf":
        m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=30.0, min_near=0.2)
    else:
        m = network.PaletteNetwork(renderer.default_opt(), bound=2, cuda_ray=True, density_scale=30.0, min_near=0.2)
    scene.seed_field_(m, 5)
    m = m.to(cuda).eval()
    put_scene(m, cuda)
    m.count_rendered = True
    m.march_mode, m.fused_field = "native", True
    H, W = 40, 56
    pose = torch.from_numpy(scene.lookat_pose())[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(H, W), H, W)
    ro, rd = ro.to(cuda), rd.to(cuda)
    m._fused = (NeRFFieldFused if model_kind == "nerf" else PaletteFieldFused)(m)
    kw = dict(perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    with torch.no_grad():
        base = m.render(ro, rd, **kw)
        outs = []
        for order in (tile_ray_order(torch.arange(H * W), W, 8), torch.randperm(H * W, generator=torch.Generator().manual_seed(1)).to(torch.int32)):
            m._fused.ray_order = order.to(cuda)
            outs.append(m.render(ro, rd, **kw))
    assert int(base["rendered"].item()) > 1000
    for o in outs:
        assert int(o["rendered"].item()) == int(base["rendered"].item())
        for k, v in base.items():
            if torch.is_tensor(v) and v.dtype.is_floating_point and v.numel() > 1:
                assert torch.equal(torch.nan_to_num(o[k], nan=-7.0), torch.nan_to_num(v, nan=-7.0)), k


def test_training_converges_on_a_synthetic_scene(cuda):
    """End to end: 400 reference-style training steps (profiles/train_convergence.py) must lift the held-out PSNR from ~9 dB
    to well above 30 dB (45 dB after 1500 steps at 200x200, profiles/r01_train_convergence.txt)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "train_convergence.py")
    spec = importlib.util.spec_from_file_location("train_convergence", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    log = mod.main(["--steps", "500", "--res", "96", "--views", "12", "--rays", "2048"])
    assert log[0][1] < 20.0 and log[-1][1] > 30.0 and log[-1][1] - log[0][1] > 15.0, log


@pytest.mark.parametrize("pred_clip", [False, True])
def test_palette_native_loop_pair_table_is_bit_identical(cuda, pred_clip):
    """The interleaved (encoder, encoder_palette) table of the native PaletteNeRF loop -- with a clip head the three-table copy with 32-byte
    rows -- must leave every output bit-identical to the separate lookups, and must follow in-place updates of a table."""
    from palettenerf_amd.fused import PaletteFieldFused
    m = network.PaletteNetwork(renderer.default_opt(pred_clip=pred_clip), bound=2, cuda_ray=True, density_scale=30.0, min_near=0.2)
    scene.seed_field_(m, 9)
    m = m.to(cuda).eval()
    put_scene(m, cuda)
    m.count_rendered = True
    m.march_mode, m.fused_field = "native", True
    m._fused = PaletteFieldFused(m)
    pose = torch.from_numpy(scene.lookat_pose())[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(48, 40), 48, 40)
    ro, rd = ro.to(cuda), rd.to(cuda)
    kw = dict(perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4, gui_mode=False)

    def both():
        outs = []
        for flag in (False, True):
            m._fused.interleave_tables = flag
            with torch.no_grad():
                outs.append(m.render(ro, rd, **kw))
        a, b = outs
        assert int(a["rendered"].item()) == int(b["rendered"].item()) > 500
        for k, v in a.items():
            if torch.is_tensor(v) and v.dtype.is_floating_point and v.numel() > 1:
                assert torch.equal(torch.nan_to_num(b[k], nan=-7.0), torch.nan_to_num(v, nan=-7.0)), k
        return a
    first = both()
    assert (getattr(m._fused, "_triple", None) is not None) == pred_clip
    with torch.no_grad():
        m.encoder_palette.embeddings.mul_(0.5)      # in-place update: the interleaved copy must be rebuilt
        if pred_clip:
            m.encoder_clip.embeddings.mul_(0.5)
    second = both()
    assert not torch.equal(first["basis_rgb"], second["basis_rgb"])
    if pred_clip:
        assert float(first["clip_feat"].abs().max()) > 0 and not torch.equal(first["clip_feat"], second["clip_feat"])


@pytest.mark.parametrize("precision", [0, 1])
def test_nerf_composite_fusion_is_bit_identical(cuda, precision):
    """NeRF native loop with every iteration composited inside the field kernel (composite_fusion 2, the default: no composite launch at all;
    rays of 2 ... 8 rows walked through wave shuffles, wave tiles of whole rays, survivor counts by atomics) and with the one-sample-per-ray
    iterations only (1) against the same loop with the separate composite launch (0): image, depth, weights and the sample count
    bit-identical.  An opaque field (most iterations have n_step 1, rays die by T_thresh or by leaving the solid) and translucent ones
    (n_step climbs through 2, 3, ... 8); 100 x 90 rays: the last chunk of the alive list is partial."""
    from palettenerf_amd import _lib
    from palettenerf_amd.fused import NeRFFieldFused
    lib = _lib.load()
    pose = torch.from_numpy(scene.lookat_pose())[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(100, 90), 100, 90)
    ro, rd = ro.to(cuda), rd.to(cuda)
    try:
        for density in (100.0, 0.5, 0.05):
            m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=density, min_near=0.2)
            scene.seed_field_(m, 23)
            m = m.to(cuda).eval()
            put_scene(m, cuda)
            m.count_rendered = True
            m.march_mode, m.fused_field = "native", True
            m._fused = NeRFFieldFused(m)
            m._fused.precision = precision
            outs = []
            for flag in (0, 1, 2):
                assert lib.pnr_set_option(b"composite_fusion", flag) == 0
                with torch.no_grad():
                    outs.append(m.render(ro, rd, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4))
            b = outs[0]
            for a in outs[1:]:
                assert int(a["rendered"].item()) == int(b["rendered"].item()) > 1000 and a["iterations"] == b["iterations"]
                if density < 1:
                    assert a["iterations"] > 30
                for k in ("image", "depth", "weights_sum"):
                    assert torch.equal(torch.nan_to_num(b[k], nan=-7.0), torch.nan_to_num(a[k], nan=-7.0)), (density, k)
    finally:
        lib.pnr_set_option(b"composite_fusion", 2)


def test_palette_aux_fusion_is_bit_identical(cuda):
    """PaletteNeRF native loop with the whole compositing step inside the field kernel (the default: aux rows and the ray state, any number of
    samples per ray, wave tiles of whole rays; no composite launch), with the aux composite only inside it (composite_fusion 1: 1, 2, 4, 8
    samples per ray) and with the separate composite launch doing everything (aux_fusion 0): every map bit-identical.  Three densities so
    that the frame goes through n_step 1 ... 8 and through the T_thresh early exits; with and without a clip head (52- and 36-float rows)."""
    from palettenerf_amd import _lib
    from palettenerf_amd.fused import PaletteFieldFused
    lib = _lib.load()
    pose = torch.from_numpy(scene.lookat_pose())[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(64, 56), 64, 56)
    ro, rd = ro.to(cuda), rd.to(cuda)
    try:
        for density, pred_clip in ((40.0, False), (0.5, False), (0.05, False), (0.5, True)):
            m = network.PaletteNetwork(renderer.default_opt(pred_clip=pred_clip), bound=2, cuda_ray=True, density_scale=density, min_near=0.2)
            scene.seed_field_(m, 21)
            m = m.to(cuda).eval()
            put_scene(m, cuda)
            m.count_rendered = True
            m.march_mode, m.fused_field = "native", True
            m._fused = PaletteFieldFused(m)
            outs = []
            for aux, comp in ((0, 2), (1, 2), (1, 1)):
                assert lib.pnr_set_option(b"aux_fusion", aux) == 0 and lib.pnr_set_option(b"composite_fusion", comp) == 0
                with torch.no_grad():
                    outs.append(m.render(ro, rd, perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4, gui_mode=False))
            b = outs[0]
            for a in outs[1:]:
                assert int(a["rendered"].item()) == int(b["rendered"].item()) > 1000 and a["iterations"] == b["iterations"]
                if density < 1:
                    assert a["iterations"] > 30          # translucent: the schedule reaches 8 samples per ray
                for k, v in a.items():
                    if torch.is_tensor(v) and v.dtype.is_floating_point and v.numel() > 1:
                        assert torch.equal(torch.nan_to_num(b[k], nan=-7.0), torch.nan_to_num(v, nan=-7.0)), (density, pred_clip, k)
    finally:
        lib.pnr_set_option(b"aux_fusion", 1); lib.pnr_set_option(b"composite_fusion", 2)
    assert lib.pnr_set_option(b"no_such_option", 1) != 0


@pytest.mark.parametrize("model_kind", ["nerf", "palette"])
def test_native_loop_half_tables_track_the_fp32_frame(cuda, model_kind):
    """table_half: the native loop looks the hash tables up as fp16 with the reference's half interpolation (its --fp16 mode; the lookup
    kernel shares corner_accumulate<__half> with pnr_grid_encode_forward, which is bit-exact against the oracle).  The frame must
    stay within fp16 rounding of the fp32-table frame: image 4e-3, sample count 2 %."""
    from palettenerf_amd.fused import NeRFFieldFused, PaletteFieldFused
    if model_kind == "nerf":
        m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=30.0, min_near=0.2)
    else:
        m = network.PaletteNetwork(renderer.default_opt(), bound=2, cuda_ray=True, density_scale=30.0, min_near=0.2)
    scene.seed_field_(m, 13)
    m = m.to(cuda).eval()
    put_scene(m, cuda)
    m.count_rendered = True
    m.march_mode, m.fused_field = "native", True
    m._fused = (NeRFFieldFused if model_kind == "nerf" else PaletteFieldFused)(m)
    pose = torch.from_numpy(scene.lookat_pose())[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(64, 64), 64, 64)
    ro, rd = ro.to(cuda), rd.to(cuda)
    kw = dict(perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    outs = []
    for half in (False, True):
        m._fused.table_half = half
        with torch.no_grad():
            outs.append(m.render(ro, rd, **kw))
    a, b = outs
    na, nb_ = int(a["rendered"].item()), int(b["rendered"].item())
    assert na > 1000 and abs(na - nb_) < 0.02 * na
    diff = (a["image"] - b["image"]).abs()
    assert float(diff.max()) < 2e-2 and float(diff.mean()) < 4e-3 and float(diff.max()) > 0.0, (float(diff.max()), float(diff.mean()))


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("case", ["a", "b"])
def test_uniform_sampling_path_on_the_hip_ops(cuda, golden_dir, case, fused):
    """BASELINE configs[0] on the GPU: NeRFRenderer.run (no occupancy grid) over the HIP near/far, hash-grid and SH operators against the
    frame the reference's own run() produced (tests/golden/gen_golden.py run); fp32 colour tolerance 1e-4."""
    g = load(golden_dir, f"run_nerf_{case}")
    m = network.NeRFNetwork(bound=2, cuda_ray=False, density_scale=float(g["density_scale"]), min_near=0.2)
    scene.seed_field_(m, int(g["seed"]))
    m = m.to(cuda).eval()
    m.fused_field = fused   # True: sigma and rgb of every point from the fused MFMA field kernel, unmasked points zeroed afterwards
    ro, rd = frame_rays(g, cuda)
    kw = dict(num_steps=int(g["num_steps"]), upsample_steps=int(g["upsample_steps"]), perturb=False)
    with torch.no_grad():
        r = m.render(ro, rd, staged=True, max_ray_batch=4096, **kw)
    close(r["image"], g["image"], what="image")
    close(r["weights_sum"], g["weights_sum"], what="weights_sum")
    close(r["depth"], g["depth"], what="depth")


@pytest.mark.parametrize("model_kind", ["nerf", "palette"])
def test_native_loop_under_fp16_autocast_uses_half_tables(cuda, model_kind):
    """The reference's -O mode (fp16 autocast): the native loop then looks the tables up as fp16 with the reference's half interpolation and keeps
    the field on its fp32-accurate path.  Same frame, bit for bit, as table_half=True outside autocast; fp32 outputs."""
    from palettenerf_amd.fused import NeRFFieldFused, PaletteFieldFused
    if model_kind == "nerf":
        m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=30.0, min_near=0.2)
    else:
        m = network.PaletteNetwork(renderer.default_opt(), bound=2, cuda_ray=True, density_scale=30.0, min_near=0.2)
    scene.seed_field_(m, 17)
    m = m.to(cuda).eval()
    put_scene(m, cuda)
    m.count_rendered = True
    m.march_mode, m.fused_field = "native", True
    m._fused = (NeRFFieldFused if model_kind == "nerf" else PaletteFieldFused)(m)
    pose = torch.from_numpy(scene.lookat_pose())[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(48, 48), 48, 48)
    ro, rd = ro.to(cuda), rd.to(cuda)
    kw = dict(perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a = m.render(ro, rd, **kw)
    assert m._fused.table_half is False          # restored
    m._fused.table_half = True
    with torch.no_grad():
        b = m.render(ro, rd, **kw)
    assert a["image"].dtype == torch.float32 and int(a["rendered"].item()) == int(b["rendered"].item()) > 500
    for k in ("image", "depth", "weights_sum"):
        assert torch.equal(torch.nan_to_num(a[k], nan=-7.0), torch.nan_to_num(b[k], nan=-7.0)), k


def test_fused_blobs_follow_parameter_updates(cuda, golden_dir):
    """The packed MFMA weights, interleaved tables and host-side parameter copies are caches of the parameters: optimizer-style in-place
    updates, load_state_dict and initialize_palette must all show up in the next native frame; so must writes through `.data` (an EMA swap,
    nerf/utils.py:829-839), which torch's version counters do not see (fused._SourceWatch)."""
    g = load(golden_dir, "frame_palette_a")
    opt = renderer.default_opt()
    m = network.PaletteNetwork(opt, bound=2, cuda_ray=True, density_scale=float(g["density_scale"]), min_near=0.2)
    scene.seed_field_(m, int(g["seed"]))
    m = m.to(cuda).eval()
    put_scene(m, cuda)
    m.march_mode, m.fused_field = "native", True
    ro, rd = frame_rays(g, cuda)
    kw = dict(dt_gamma=float(g["dt_gamma"]), perturb=False, max_steps=1024, T_thresh=1e-4, gui_mode=False)

    def both():
        with torch.no_grad():
            n = m.render(ro, rd, **kw)["image"]
            m.march_mode, m.fused_field = "compat", False
            c = m.render(ro, rd, **kw)["image"]
            m.march_mode, m.fused_field = "native", True
        return n, c

    n0, c0 = both()
    close(n0, g["image"], what="before")
    # 1. in-place update of weights, a table, the bias and the palette (what an optimizer step does)
    with torch.no_grad():
        m.color_net[2].weight.mul_(0.5)
        m.encoder_palette.embeddings.mul_(-1.0)
        m.offsets_radiance_net.bias.add_(0.3)
        m.basis_color.mul_(0.5)
    n1, c1 = both()
    assert float((n1 - n0).abs().max()) > 1e-2 and float((n1 - c1).abs().max()) < COLOUR_TOL
    # 2. a write through .data moves neither identity nor version (torch_ema's copy_to / restore, nerf/utils.py:829-839): the frame loop's
    #    per-frame checksum of the blobs' sources notices it, rebuilds and renders again -- no invalidate_fused_caches() call needed
    m.color_net[1].weight.data.mul_(-1.0)
    with pytest.warns(UserWarning, match="rewritten behind torch's version counters"):
        n2, c2 = both()
    assert float((n2 - n1).abs().max()) > 1e-3 and float((n2 - c2).abs().max()) < COLOUR_TOL
    m.encoder_palette.embeddings.data.copy_(m.encoder_palette.embeddings.data.flip(0))       # a table (sampled checksum), as an EMA swap rewrites it
    m.basis_color.data.add_(0.05)
    with pytest.warns(UserWarning, match="rewritten behind torch's version counters"):
        n2b, c2b = both()
    assert float((n2b - n2).abs().max()) > 1e-3 and float((n2b - c2b).abs().max()) < COLOUR_TOL
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")               # nothing changed: no second render, no warning
        n2c, _ = both()
    assert torch.equal(n2c, n2b)
    m.color_net[1].weight.data.mul_(-1.0)            # the documented way still works, and without the double render
    m.invalidate_fused_caches()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        n2d, c2d = both()
    assert float((n2d - c2d).abs().max()) < COLOUR_TOL
    # 3. load_state_dict (same Parameter objects, new values) and a second initialize_palette (a NEW Parameter object at version 0)
    fresh = network.PaletteNetwork(opt, bound=2, cuda_ray=True, density_scale=float(g["density_scale"]), min_near=0.2)
    scene.seed_field_(fresh, int(g["seed"]))
    sd = {k: v for k, v in fresh.state_dict().items() if not k.startswith("density_")}
    m.load_state_dict(sd, strict=False)
    n3, c3 = both()
    close(n3, g["image"], what="after load_state_dict")
    m.initialize_palette([[0.9, 0.1, 0.1], [0.1, 0.9, 0.1], [0.1, 0.1, 0.9], [0.5, 0.5, 0.5]])
    n4, c4 = both()
    assert float((n4 - n3).abs().max()) > 1e-2 and float((n4 - c4).abs().max()) < COLOUR_TOL
    m.opt.color_space = "linear"   # main_palette.py --color_space linear: the extracted sRGB palette is linearised (palette/renderer.py:256-259)
    m.initialize_palette([[0.9, 0.1, 0.1], [0.1, 0.9, 0.1], [0.1, 0.1, 0.9], [0.5, 0.5, 0.5]])
    want = torch.tensor([0.9, 0.1, 0.5])
    want = torch.where(want < 0.04045, want / 12.92, ((want + 0.055) / 1.055) ** 2.4)
    assert torch.allclose(m.basis_color.detach().cpu()[[0, 0, 3], [0, 1, 0]], want, atol=1e-7)
    assert torch.equal(m.basis_color_origin, m.basis_color.detach())


@pytest.mark.parametrize("case", ["style_a", "style_b", "nb6", "nb8"])
def test_palette_stylizer_edit_and_many_basis_frames_in_every_mode(cuda, golden_dir, case):
    """Fixtures from the reference's own Stylizer / RegionEdit classes and PaletteNetwork with 4, 6 and 8 palette bases
    (tests/golden/gen_golden.py:gen_palette_extra).  Both editing heads run inside the fused field kernel's epilogue, in the host-driven
    loop (fused) and in the device-driven loop (native), in the split-fp16 and the exact-fp32 matrix path."""
    from tests.test_host_logic import _extra_model, set_extra_edit, set_extra_stylizer
    from palettenerf_amd.fused import PaletteFieldFused
    g = load(golden_dir, f"frame_palette_{case}")
    opt, m = _extra_model(g)
    m = m.to(cuda).eval()
    put_scene(m, cuda)
    ro, rd = frame_rays(g, cuda)
    kw = dict(dt_gamma=float(g["dt_gamma"]), perturb=False, max_steps=1024, T_thresh=1e-4)
    maps = ("image", "weights_sum", "clip_feat", "direct_rgb", "view_dep_rgb", "basis_rgb", "unscaled_basis_rgb", "basis_acc")
    for mode in ("compat", "fused", "native", "native_fp32"):
        m.march_mode = {"fused": "device", "native_fp32": "native"}.get(mode, mode)
        m.fused_field = mode != "compat"
        if m.fused_field:
            m._fused = PaletteFieldFused(m)
            m._fused.precision = 0 if mode == "native_fp32" else 1
        with torch.no_grad():
            r = m.render(ro, rd, gui_mode=False, **kw)
            for k in maps:
                close(r[k], g[k], what=f"{mode}:{k}")
            close(r["depth"], g["depth"], tol=DEPTH_TOL, what=f"{mode}:depth")
            set_extra_stylizer(m, opt, g, cuda)
            close(m.render(ro, rd, gui_mode=True, **kw)["image"], g["style_image"], what=f"{mode}:style_image")
            if m.fused_field:
                with pytest.raises(RuntimeError, match="gui_mode"):
                    m.render(ro, rd, gui_mode=False, **kw)
            m.stylizer = None
            set_extra_edit(m, opt, cuda)
            e = m.render(ro, rd, gui_mode=False, **kw)
            close(e["image"], g["edit_image"], tol=EDIT_TOL, what=f"{mode}:edit_image")
            close(e["basis_rgb"], g["edit_basis_rgb"], tol=EDIT_TOL, what=f"{mode}:edit_basis_rgb")
            m.edit.weight_mode = True
            close(m.render(ro, rd, gui_mode=True, **kw)["image"], g["edit_weight_image"], what=f"{mode}:edit_weight_image")
            m.edit = None


def test_native_loop_renders_the_round1_edit_fixture(cuda, golden_dir):
    """frame_palette_{a,b}.npz: edit_image (RegionEdit with a spatial window only, gui_mode) now also through the device-driven loop."""
    for case in ("a", "b"):
        g = load(golden_dir, f"frame_palette_{case}")
        opt = renderer.default_opt(pred_clip=bool(g["pred_clip"]))
        m = network.PaletteNetwork(opt, bound=2, cuda_ray=True, density_scale=float(g["density_scale"]), min_near=0.2)
        scene.seed_field_(m, int(g["seed"]))
        m = m.to(cuda).eval()
        put_scene(m, cuda)
        ro, rd = frame_rays(g, cuda)
        m.march_mode, m.fused_field = "native", True
        m.edit = renderer.RegionEdit(opt)
        m.edit.update_cent(mean_xyz=torch.tensor([0.1, 0.0, -0.2], device=cuda))
        m.edit.update_std(std_xyz=0.5)
        m.edit.update_delta_hsv(m.basis_color.data.clamp(0, 1), (m.basis_color.data * 0.6 + 0.2).flip(0).clamp(0, 1))
        with torch.no_grad():
            r = m.render(ro, rd, dt_gamma=float(g["dt_gamma"]), perturb=False, max_steps=1024, T_thresh=1e-4, gui_mode=True)
        close(r["image"], g["edit_image"], tol=EDIT_TOL, what=f"edit_image {case}")
        assert "iterations" in r                                                         # it really took the device-driven loop


@pytest.mark.parametrize("kind,case", [("nerf", "a"), ("nerf", "b"), ("palette", "a"), ("palette", "b")])
def test_checkpoint_file_to_native_render(cuda, golden_dir, tmp_path, kind, case):
    """f3: a `.pth` in the reference trainer's layout (nerf/utils.py:1083-1143) -> a FRESH model -> the device-driven loop.  The fused
    kernels' blobs (MFMA weight blob, interleaved pair / triple table, host-side palette and bias) are all derived from what the checkpoint
    loaded; the frame must be the fixture the reference's own renderer produced for those weights."""
    from palettenerf_amd import checkpoint
    g = load(golden_dir, f"frame_{kind}_{case}")

    def build():
        if kind == "nerf":
            return network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=float(g["density_scale"]), min_near=0.2)
        return network.PaletteNetwork(renderer.default_opt(pred_clip=bool(g["pred_clip"])), bound=2, cuda_ray=True, density_scale=float(g["density_scale"]), min_near=0.2)

    src = build()
    scene.seed_field_(src, int(g["seed"]))
    src = src.to(cuda)
    put_scene(src, cuda)
    src.mean_count, src.mean_density = 1234, 0.5
    path = checkpoint.save_model(src, str(tmp_path / "ngp_ep0007.pth"), epoch=7, global_step=700)
    del src

    m = build().to(cuda).eval()                     # default initialisation: tables U(-1e-4, 1e-4), empty occupancy
    m.march_mode, m.fused_field = "native", True
    ro, rd = frame_rays(g, cuda)
    kw = dict(dt_gamma=float(g["dt_gamma"]), perturb=False, max_steps=1024, T_thresh=1e-4)
    if kind == "palette":
        kw["gui_mode"] = False
    with torch.no_grad():
        before = m.render(ro, rd, **kw)             # packs the blobs of the UNTRAINED weights (and finds nothing to march: empty bitfield)
    assert float(before["weights_sum"].abs().max()) == 0.0
    fused = m._fused
    blob_before = fused.packed.clone()
    info = checkpoint.load_model(m, path, map_location="cpu")
    assert info["missing"] == [] and info["unexpected"] == [] and info["epoch"] == 7 and m.mean_count == 1234
    with torch.no_grad():
        r = m.render(ro, rd, **kw)
    assert m._fused is fused and not torch.equal(fused.packed, blob_before)            # same object, blob rebuilt from the loaded weights
    close(r["image"], g["image"], what="image")
    close(r["weights_sum"], g["weights_sum"], what="weights_sum")
    close(r["depth"], g["depth"], tol=DEPTH_TOL, what="depth")
    if kind == "palette":
        for k in ("clip_feat", "direct_rgb", "view_dep_rgb", "basis_rgb", "unscaled_basis_rgb", "basis_acc"):
            close(r[k], g[k], what=k)
        table = fused._triple if fused.pred_clip else fused._pair                      # the interleaved copy holds the LOADED tables
        assert torch.equal(table[:, 0:2], m.encoder.embeddings.detach()) and torch.equal(table[:, 2:4], m.encoder_palette.embeddings.detach())
    # a 'best' checkpoint (no density_grid, utils.py:1135) loads non-strictly and leaves the occupancy alone
    best = checkpoint.save_model(m, str(tmp_path / "ngp.pth"), best=True)
    info = checkpoint.load_model(build().to(cuda), best)
    assert info["missing"] == ["density_grid"]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["nerf", "palette"])
def test_frames_in_flight_equal_frames_rendered_one_by_one(cuda, kind):
    """palettenerf_amd.pipeline.FramesInFlight: three frames in flight (three host threads, each with its own fused-field object, workspace
    and stream on the SAME weights) give, pose by pose, bit for bit the images, depths and sample counts of the frames rendered one after
    another -- the per-thread / per-device state of pnr_*_render_frame and the shared read-only caches hold under concurrency."""
    from palettenerf_amd import network, renderer, scene
    from palettenerf_amd.fused import NeRFFieldFused, PaletteFieldFused
    from palettenerf_amd.pipeline import FramesInFlight
    if kind == "nerf":
        m = network.NeRFNetwork(bound=2, cuda_ray=True)
    else:
        m = network.PaletteNetwork(renderer.default_opt(), bound=2, cuda_ray=True)
    scene.seed_field_(m, 0)
    m = m.to(cuda).eval()
    m.density_grid.copy_(torch.from_numpy(scene.brick_density_grid()).to(cuda))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    m.march_mode, m.fused_field, m.count_rendered = "native", True, True
    m._fused = NeRFFieldFused(m) if kind == "nerf" else PaletteFieldFused(m)
    H = W = 160
    intr = scene.intrinsics_from_fov(H, W)
    rays = []
    for i in range(7):
        pose = torch.from_numpy(scene.lookat_pose(azimuth_deg=20.0 + 17.0 * i))[None]
        ro, rd = scene.get_rays(pose, intr, H, W)
        rays.append((ro.to(cuda), rd.to(cuda)))
    kw = dict(perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    if kind == "palette":
        kw["gui_mode"] = False
    keys = ("image", "depth", "weights_sum") + (("basis_rgb", "basis_acc", "view_dep_rgb") if kind == "palette" else ())
    with torch.no_grad():
        one_by_one = [m.render(ro, rd, **kw) for ro, rd in rays]
    want = [{k: r[k].clone() for k in keys} | {"rendered": int(r["rendered"].sum())} for r in one_by_one]
    for shared in (False, True):   # a stream per handle (kernels of different frames overlap) / one stream for all (frames back to back, the host's gap hidden)
        fif = FramesInFlight(m, 3, shared_stream=shared)
        for _ in range(2):   # twice: the second pass runs on warm handles (each thread's iteration prediction comes from a different pose)
            got = fif.render(lambda i: rays[i], len(rays), **kw)
            for i, (g, w) in enumerate(zip(got, want)):
                assert int(g["rendered"].sum()) == w["rendered"], i
                for k in keys:
                    a, b = g[k].cpu().numpy(), w[k].cpu().numpy()
                    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
                    np.testing.assert_array_equal(np.nan_to_num(a), np.nan_to_num(b), err_msg=f"frame {i} {k} shared_stream={shared}")
        fif.close()


@pytest.mark.gpu
def test_render_path_equals_the_reference_test_loop_arithmetic(cuda):
    """pipeline.render_path (Trainer.test's inner loop, nerf/utils.py:704-731, with two poses in flight) against the same loop written out as the
    reference does it: get_rays -> render -> `(pred * 255).astype(np.uint8)` on the host.  Bytes are equal wherever the float is finite (the
    reference's cast of the NaN depth of rays that miss the scene, quirk 7, is platform-defined)."""
    from palettenerf_amd.fused import NeRFFieldFused
    from palettenerf_amd import rays as prays
    from palettenerf_amd.pipeline import render_path
    m = network.NeRFNetwork(bound=2, cuda_ray=True)
    scene.seed_field_(m, 0)
    m = m.to(cuda).eval()
    m.density_grid.copy_(torch.from_numpy(scene.brick_density_grid()).to(cuda))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    m.march_mode, m.fused_field = "native", True
    m._fused = NeRFFieldFused(m)
    H, W = 96, 128
    intr = scene.intrinsics_from_fov(H, W)
    poses = np.stack([scene.lookat_pose(azimuth_deg=30.0 + 40.0 * i) for i in range(5)])
    kw = dict(perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4, bg_color=1)
    rgb, dep = render_path(m, poses, intr, H, W, frames_in_flight=2, **kw)
    assert rgb.shape == (5, H, W, 3) and rgb.dtype == np.uint8 and dep.shape == (5, H, W) and dep.dtype == np.uint8
    with torch.no_grad():
        for i in range(5):
            ro, rd = prays.rays_from_indices(torch.from_numpy(poses[i:i + 1]).to(cuda), intr, H, W, None)   # the device ray generator render_path uses
            r = m.render(ro, rd, **kw)
            pred = r["image"].reshape(H, W, 3).cpu().numpy()
            pdep = r["depth"].reshape(H, W).cpu().numpy()
            np.testing.assert_array_equal(rgb[i], (pred * 255).astype(np.uint8))
            ok = np.isfinite(pdep)
            assert ok.mean() > 0.2
            np.testing.assert_array_equal(dep[i][ok], (pdep[ok] * 255).astype(np.uint8))


@pytest.mark.gpu
def test_palette_frame_f16x2_is_inside_the_colour_contract(cuda):
    """PNR_FIELD_F16X2 on the PaletteNeRF frame loop (opt-in; the specialised 4-basis kernel with the colour heads' activations rounded once to fp16,
    sigma_net in the split form): every map within 1e-4 of the split-fp16 frame (north-star colour contract), PSNR above 85 dB, exactly the same
    samples and accumulated alpha.  With an edit head or another basis count the library runs the split form: bit-identical."""
    from palettenerf_amd.fused import PaletteFieldFused
    m = network.PaletteNetwork(renderer.default_opt(), bound=2, cuda_ray=True, density_scale=100.0)
    scene.seed_field_(m, 0)
    m = m.to(cuda).eval()
    m.density_grid.copy_(torch.from_numpy(scene.brick_density_grid()).to(cuda))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    m.march_mode, m.fused_field, m.count_rendered = "native", True, True
    m._fused = PaletteFieldFused(m)
    H = W = 160
    ro, rd = scene.get_rays(torch.from_numpy(scene.lookat_pose())[None], scene.intrinsics_from_fov(H, W), H, W)
    ro, rd = ro.to(cuda), rd.to(cuda)
    kw = dict(perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4, gui_mode=False, bg_color=1)
    keys = ("image", "direct_rgb", "view_dep_rgb", "basis_rgb", "unscaled_basis_rgb", "basis_acc", "weights_sum")
    with torch.no_grad():
        ref = m.render(ro, rd, **kw)
        m._fused.precision = 2
        fast = m.render(ro, rd, **kw)
    assert int(fast["rendered"].sum()) == int(ref["rendered"].sum()) and torch.equal(fast["weights_sum"], ref["weights_sum"])
    worst = 0.0
    for k in keys:
        worst = max(worst, float((fast[k] - ref[k]).abs().max()))
    assert 1e-8 < worst < 1e-4, worst      # really the rounded form, and inside the contract
    assert scene.psnr(fast["image"].cpu(), ref["image"].cpu()) > 85.0
    # an edit head: the library falls back to the split form
    m.edit = renderer.RegionEdit(m.opt)
    m.edit.update_cent(mean_xyz=torch.tensor([0.2, 0.1, -0.1], device=cuda))
    m.edit.update_std(std_xyz=0.3)
    m.edit.update_delta_hsv(m.basis_color.data.clamp(0, 1), (m.basis_color.data * 0.5 + 0.3).flip(0).clamp(0, 1))
    with torch.no_grad():
        e2 = m.render(ro, rd, **kw)
        m._fused.precision = 1
        e1 = m.render(ro, rd, **kw)
    assert torch.equal(e1["image"], e2["image"])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["nerf", "palette"])
@pytest.mark.parametrize("sorted_rays", [True, False])
def test_frame_call_computes_near_far_and_epilogue_itself(cuda, kind, sorted_rays):
    """pnr_{nerf,palette}_frame_args::aabb / finish / depth_raw (round 5): the frame call's first launch computes near / far (near_far_from_aabb's
    arithmetic, raymarching.cu:95-148) and its last launch applies run_cuda's epilogue (nerf/renderer.py:382-384, palette/renderer.py:520-540:
    background blend of image and direct_rgb, depth normalisation, depth_origin).  Bit for bit what the operator call and the torch expressions give
    around a frame call that is handed nears / fars and asked for raw accumulations -- with a ray order (gathered copies) and without (in place)."""
    from palettenerf_amd.fused import NeRFFieldFused, PaletteFieldFused, tile_ray_order
    if kind == "nerf":
        m = network.NeRFNetwork(bound=2, cuda_ray=True, min_near=0.05)
    else:
        m = network.PaletteNetwork(renderer.default_opt(), bound=2, cuda_ray=True, density_scale=100.0, min_near=0.05)
    scene.seed_field_(m, 0)
    m = m.to(cuda).eval()
    m.density_grid.copy_(torch.from_numpy(scene.brick_density_grid()).to(cuda))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    f = (NeRFFieldFused if kind == "nerf" else PaletteFieldFused)(m)
    H, W = 96, 128
    ro, rd = scene.get_rays(torch.from_numpy(scene.lookat_pose(azimuth_deg=25.0))[None], scene.intrinsics_from_fov(H, W), H, W)
    ro, rd = ro[0].to(cuda).contiguous(), rd[0].to(cuda).contiguous()
    N = H * W
    ro[: N // 8, 0] += 10.0       # an eighth of the rays pass beside the box ...
    ro[N // 8: N // 4] *= 0.1     # ... and another starts inside it (near clamped to min_near)
    f.ray_order = tile_ray_order(torch.arange(N), W, 8).to(cuda) if sorted_rays else None
    nears, fars = raymarching.near_far_from_aabb(ro, rd, m.aabb_infer, m.min_near)
    assert bool((nears == m.min_near).any()) and bool((nears > m.min_near).any()) and bool((fars < 3.0e38).any()) and bool((fars > 3.0e38).any())
    bg = torch.rand(N, 3, device=cuda)
    for bg_color in (1, (0.2, 0.5, 0.9), bg):
        raw = f.render_frame(ro, rd, nears, fars, 0.0, 1024, 1e-4)
        done = f.render_frame(ro, rd, None, None, 0.0, 1024, 1e-4, bg_color=bg_color, aabb=m.aabb_infer, min_near=m.min_near)
        st_raw, st = raw[-1], done[-1]
        assert st["finished"] and not st_raw.get("finished") and st["rendered"] == st_raw["rendered"] > 10_000
        assert torch.equal(st["nears"], nears) and torch.equal(st["fars"], fars)
        ws, depth, image = raw[0], raw[1], raw[2]
        bgt = bg_color if torch.is_tensor(bg_color) else torch.tensor([bg_color] * 3 if isinstance(bg_color, int) else bg_color, dtype=torch.float32, device=cuda)
        assert torch.equal(done[0], ws)
        assert torch.equal(done[2], image + (1 - ws).unsqueeze(-1) * bgt)
        want_depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        assert torch.equal(done[1].isnan(), want_depth.isnan())      # rays that miss the box: (FLT_MAX - FLT_MAX) in the denominator, quirk 7
        ok = ~want_depth.isnan()
        assert torch.equal(done[1][ok], want_depth[ok])
        if kind == "palette":
            assert torch.equal(st["depth_raw"], depth)
            assert torch.equal(done[3][:, 3:], raw[3][:, 3:])
            assert torch.equal(done[3][:, 0:3], raw[3][:, 0:3] + (1 - ws).unsqueeze(-1) * bgt)


@pytest.mark.gpu
def test_kept_frame_arguments_follow_every_setting(cuda):
    """The frame calls keep their argument struct between frames (fused.py: `_frame_plan`).  Everything the struct is filled from must be part of its key:
    after each change of a setting, the same object's next frame equals the frame of a fresh PaletteFieldFused built after the change."""
    from palettenerf_amd.fused import PaletteFieldFused, tile_ray_order
    m = network.PaletteNetwork(renderer.default_opt(), bound=2, cuda_ray=True, density_scale=100.0)
    scene.seed_field_(m, 0)
    m = m.to(cuda).eval()
    m.density_grid.copy_(torch.from_numpy(scene.brick_density_grid()).to(cuda))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    H, W = 64, 96
    ro, rd = scene.get_rays(torch.from_numpy(scene.lookat_pose(azimuth_deg=40.0))[None], scene.intrinsics_from_fov(H, W), H, W)
    ro, rd = ro[0].to(cuda).contiguous(), rd[0].to(cuda).contiguous()
    N = H * W
    kept = PaletteFieldFused(m)

    def frames_agree(what):
        fresh = PaletteFieldFused(m)
        fresh.ray_order, fresh.precision, fresh.table_half = kept.ray_order if hasattr(kept, "ray_order") else None, kept.precision, kept.table_half
        a = kept.render_frame(ro, rd, None, None, 0.0, 1024, 1e-4, bg_color=1, aabb=m.aabb_infer, min_near=m.min_near)
        b = fresh.render_frame(ro, rd, None, None, 0.0, 1024, 1e-4, bg_color=1, aabb=m.aabb_infer, min_near=m.min_near)
        assert a[-1]["rendered"] == b[-1]["rendered"] > 1000, what
        for x, y in zip(a[:4], b[:4]):
            assert torch.equal(x, y), what
        return a

    kept.ray_order = None
    base = frames_agree("first frame")
    frames_agree("second frame, nothing changed")
    m.view_dep_weight = 0.25
    v = frames_agree("view_dep_weight")
    assert not torch.equal(v[2], base[2])
    m.offsets_weight = 0.5
    frames_agree("offsets_weight")
    m.density_scale = 50.0
    d = frames_agree("density_scale")
    assert not torch.equal(d[0], v[0])
    kept.ray_order = tile_ray_order(torch.arange(N), W, 8).to(cuda)
    frames_agree("ray_order")
    kept.precision = 0
    frames_agree("precision fp32")
    kept.precision = 1
    with torch.no_grad():
        m.basis_color.add_(0.03)                       # an in-place torch op: the version counter moves
    frames_agree("palette changed in place")
    m.density_grid.mul_(0.0).add_(torch.from_numpy(scene.brick_density_grid()).to(cuda).roll(3, dims=1))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)       # another occupancy grid: bitfield version, mip
    frames_agree("occupancy grid")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["nerf", "palette"])
def test_kept_frame_arguments_survive_a_stand_alone_field_call_at_another_precision(cuda, kind):
    """weights that fail the STATIC fp16 bound run the stand-alone ops (`model.forward` -> `self._fused(x, d)`) on the exact fp32 path
    (effective_precision() == 0) while the frame loops keep split-fp16 with the overflow watch (frame_precision()).  The stand-alone call repacks the blob for
    ITS precision -- in place (NeRF) or into a new tensor (PaletteNeRF) -- so the frame call must not keep the blob's address or layout in its kept argument
    struct: frame, stand-alone call, frame again = the same frame bit for bit, and equal to a fresh object's frame."""
    from palettenerf_amd.fused import NeRFFieldFused, PaletteFieldFused
    if kind == "nerf":
        m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=100.0)
        cls = NeRFFieldFused
    else:
        m = network.PaletteNetwork(renderer.default_opt(), bound=2, cuda_ray=True, density_scale=100.0)
        cls = PaletteFieldFused
    scene.seed_field_(m, 3)
    m = m.to(cuda).eval()
    m.density_grid.copy_(torch.from_numpy(scene.brick_density_grid()).to(cuda))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    H, W = 48, 64
    ro, rd = scene.get_rays(torch.from_numpy(scene.lookat_pose(azimuth_deg=25.0))[None], scene.intrinsics_from_fov(H, W), H, W)
    ro, rd = ro[0].to(cuda).contiguous(), rd[0].to(cuda).contiguous()

    def pessimistic(obj):
        obj._guard_bound = lambda tmax, scales: 1.0e9      # "trained weights": the product of L1 norms is far beyond fp16, the activations are not
        return obj

    kept = pessimistic(cls(m))
    assert kept.effective_precision() == 0 and kept.frame_precision() == (1, True)
    frame = lambda obj: obj.render_frame(ro, rd, None, None, 0.0, 1024, 1e-4, bg_color=1, aabb=m.aabb_infer, min_near=m.min_near)
    first = frame(kept)
    assert first[-1]["rendered"] > 1000
    x = (torch.rand(4096, 3, device=cuda) * 2 - 1) * 1.5
    d = torch.nn.functional.normalize(torch.randn(4096, 3, device=cuda), dim=-1)
    for rounds in range(2):
        alone = kept(x, d)                                   # fp32 layout now sits in (or replaced) the blob the first frame used
        assert kept.versions[-1] == 0
        torch.empty(1 << 22, device=cuda).fill_(float("nan"))   # whatever the allocator hands out next must not be what the frame reads
        again = frame(kept)
        assert kept.versions[-1] == 1
        assert again[-1]["rendered"] == first[-1]["rendered"]
        for a, b in zip(first[:-1], again[:-1]):
            assert torch.equal(a, b)
    fresh = frame(pessimistic(cls(m)))
    for a, b in zip(first[:-1], fresh[:-1]):
        assert torch.equal(a, b)
    kept.precision = 0                                       # and the stand-alone result itself is the exact-fp32 one
    exact = kept(x, d)
    for a, b in zip(alone, exact):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["nerf", "palette"])
def test_prepare_launch_finish_frames_equal_frames_rendered_one_by_one(cuda, kind):
    """pnr_*_render_frame_submit / _finish under model.render_prepare / render_launch / render_finish (pipeline.render_queue): frame i + 1 is PREPARED while
    frame i runs -- two argument structs and two checksum rows alternate -- and every frame is bit for bit the frame render() gives, also when the submit call's
    guess of the iteration count falls short (poses far apart: the finish call then enqueues the rest) and with a ray order.  The C calls refuse a finish
    without a submit, a second submit before the finish, and a finish with another struct."""
    import ctypes
    from palettenerf_amd import _lib, network, renderer, scene
    from palettenerf_amd.fused import NeRFFieldFused, PaletteFieldFused, tile_ray_order
    from palettenerf_amd.pipeline import render_queue
    if kind == "nerf":
        m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=40.0)
    else:
        m = network.PaletteNetwork(renderer.default_opt(), bound=2, cuda_ray=True, density_scale=40.0)
    scene.seed_field_(m, 0)
    m = m.to(cuda).eval()
    m.density_grid.copy_(torch.from_numpy(scene.brick_density_grid()).to(cuda))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    m.march_mode, m.fused_field, m.count_rendered = "native", True, True
    m._fused = NeRFFieldFused(m) if kind == "nerf" else PaletteFieldFused(m)
    H, W = 96, 128
    m._fused.ray_order = tile_ray_order(torch.arange(H * W), W, 8).to(cuda)
    intr = scene.intrinsics_from_fov(H, W)
    rays = []
    for i in range(6):
        # (poses 2 and 3 look past the object: few iterations; the frame behind them needs many more than the submit call enqueues)
        pose = scene.lookat_pose_from((3.0, 1.0, 0.5), target=(6.0, 2.5, 0.5)) if i in (2, 3) else scene.lookat_pose(azimuth_deg=20.0 + 61.0 * i, elevation_deg=25.0)
        pose = torch.from_numpy(pose)[None]
        ro, rd = scene.get_rays(pose, intr, H, W)
        rays.append((ro.to(cuda), rd.to(cuda)))
    kw = dict(perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    if kind == "palette":
        kw["gui_mode"] = False
    keys = ("image", "depth", "weights_sum") + (("basis_rgb", "basis_acc", "view_dep_rgb", "direct_rgb", "depth_origin") if kind == "palette" else ())
    with torch.no_grad():
        want = []
        for ro, rd in rays:
            r = m.render(ro, rd, **kw)
            want.append({k: r[k].clone() for k in keys} | {"rendered": int(r["rendered"].sum()), "iterations": r["iterations"]})
    assert len({w["iterations"] for w in want}) > 1
    order = []
    for rounds in range(2):
        got = render_queue(m, lambda i: rays[i], len(rays), consume=lambda i, r: (order.append(i), r)[1], **kw)
        for i, (g, w) in enumerate(zip(got, want)):
            assert int(g["rendered"].sum()) == w["rendered"], i
            for k in keys:
                a, b = g[k].cpu().numpy(), w[k].cpu().numpy()
                np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
                np.testing.assert_array_equal(np.nan_to_num(a), np.nan_to_num(b), err_msg=f"frame {i} {k}")
    assert order == list(range(len(rays))) * 2
    assert max(int(g["host_looks"]) for g in got) > 1          # at least one frame's first chunk fell short: its finish call enqueued the rest
    # weights rewritten behind torch's counters BEFORE a queue run: frame 0's checksum fails in render_wait, it is rendered again (blobs rebuilt), and frame 1 --
    # prepared meanwhile against the OLD blobs -- is refused by frame_launch and prepared again inside render_launch
    m.color_net[1].weight.data.mul_(-1.0)
    with pytest.warns(UserWarning, match="rewritten behind torch's version counters"):
        got = render_queue(m, lambda i: rays[i], len(rays), **kw)
    import warnings
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")
        for i, (ro, rd) in enumerate(rays):
            r = m.render(ro, rd, **kw)
            assert int(r["rendered"].sum()) == int(got[i]["rendered"].sum())
            assert not torch.equal(torch.nan_to_num(r["image"]), torch.nan_to_num(want[i]["image"])) or int(r["rendered"].sum()) == 0
            for k in keys:
                assert torch.equal(torch.nan_to_num(r[k], nan=-7.0), torch.nan_to_num(got[i][k], nan=-7.0)), (i, k)
    m.color_net[1].weight.data.mul_(-1.0)
    m.invalidate_fused_caches()
    # the C calls' own rules
    lib = _lib.load()
    ro, rd = rays[0]
    f = m._fused
    tok_a = f.frame_prepare(ro[0], rd[0], None, None, 0.0, 1024, 1e-4, bg_color=1, aabb=m.aabb_infer, min_near=m.min_near)
    tok_b = f.frame_prepare(ro[0], rd[0], None, None, 0.0, 1024, 1e-4, bg_color=1, aabb=m.aabb_infer, min_near=m.min_near)
    sub = lib.pnr_nerf_render_frame_submit if kind == "nerf" else lib.pnr_palette_render_frame_submit
    fin = lib.pnr_nerf_render_frame_finish if kind == "nerf" else lib.pnr_palette_render_frame_finish
    arg = (lambda t: ctypes.byref(t.a)) if kind == "nerf" else (lambda t: ctypes.byref(t.p))
    assert fin(arg(tok_a), None) == -1                         # nothing submitted
    f.frame_launch(tok_a)
    assert sub(arg(tok_b), tok_a.stream) == -1                 # one submitted frame per host thread and device
    assert fin(arg(tok_b), tok_a.stream) == -1                 # not the struct that was submitted
    out = f.frame_finish(tok_a)
    assert out[-1]["rendered"] == want[0]["rendered"] and torch.equal(out[2].view(-1, 3), want[0]["image"].view(-1, 3))
    # a submitted frame its caller gave up (an exception between the two halves): the next whole-frame call drops it instead of refusing every frame from now on
    f.frame_launch(f.frame_prepare(ro[0], rd[0], None, None, 0.0, 1024, 1e-4, bg_color=1, aabb=m.aabb_infer, min_near=m.min_near))
    with torch.no_grad():
        r = m.render(rays[1][0], rays[1][1], **kw)
    assert int(r["rendered"].sum()) == want[1]["rendered"] and torch.equal(torch.nan_to_num(r["image"]), torch.nan_to_num(want[1]["image"]))
    got = render_queue(m, lambda i: rays[i], 3, **kw)
    assert all(torch.equal(torch.nan_to_num(g["image"]), torch.nan_to_num(w["image"])) for g, w in zip(got, want))
