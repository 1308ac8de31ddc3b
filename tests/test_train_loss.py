"""pnr_train_loss_* (csrc/train_loss.hip): the renderer's training epilogue + the trainer's loss (palette/utils.py:483-600, MSE criterion;
nerf/utils.py:534-556) in one launch each way, against the same arithmetic written with torch on the lazy TrainResults entries -- the
reference's own formulation (it is Python there), fp32.  Tolerances: the loss and its terms 2e-6 relative (sums reduced in another order),
gradients 1e-5 relative to the largest element."""
import ctypes

import numpy as np
import pytest
import torch

from palettenerf_amd.train_loss import RawTrain, TrainResults, TERM_NAMES, train_loss


def make_raw(N, nb, clip, bg, device, seed=0, palette=True, prefix=None):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)   # noqa: E731
    ws, depth, image = r(N) * 0.9, r(N) * 3, r(N, 3)
    nears = r(N) * 0.5
    fars = nears + 1 + r(N)
    all_map = r(N, 13 + clip + nb) if palette else None
    if bg == "scalar":
        bgc = 1
    elif bg == "rgb":
        bgc = r(3).to(device)
    else:
        bgc = r(N, 3).to(device)
    t = lambda x: None if x is None else x.to(device).requires_grad_(True)   # noqa: E731
    raw = RawTrain(t(ws), depth.to(device), t(image), t(all_map), nears.to(device), fars.to(device), bgc, prefix or (1, N), nb if palette else 0, clip if palette else 0)
    gt = r(N, 3).to(device)
    gt_clip = r(N, clip).to(device) if clip and palette else None
    gt_w = r(N, nb).to(device) if palette else None
    return raw, gt, gt_clip, gt_w


def torch_loss(res, gt, lam, gt_clip=None, gt_w=None, bc=None, bco=None):
    """palette/utils.py:483-600 / nerf/utils.py:535 on the dict entries, term by term."""
    raw = res.raw
    gt = gt.view(*raw.prefix, 3)
    loss = ((res["image"] - gt) ** 2).mean(-1)
    terms = {"loss_mse": loss.mean()}
    if raw.all_map is not None:
        am = raw.all_map
        nb, clip = raw.num_basis, raw.clip_dim
        terms["loss_sparsity"] = lam["sparsity"] * am[..., 0:1].mean()
        terms["loss_offsets"] = lam["offsets"] * am[..., 2:3].mean()
        terms["loss_view_dep"] = lam["view_dep"] * am[..., 1:2].mean()
        terms["loss_smooth"] = lam["smooth"] * am[..., 3:4].mean()
        terms["loss_palette"] = lam["palette"] * ((bc - bco) ** 2).sum(dim=-1).mean() if bc is not None else torch.zeros((), device=gt.device)
        terms["loss_weight"] = lam["weight"] * ((gt_w - am[..., 13 + clip:13 + clip + nb]) ** 2).mean() if gt_w is not None else torch.zeros((), device=gt.device)
        terms["loss_direct"] = ((res["direct_rgb"] - gt) ** 2).mean()
        terms["loss_clip_feat"] = ((am[..., 13:13 + clip] - gt_clip) ** 2).mean() if gt_clip is not None else torch.zeros((), device=gt.device)
        for k in TERM_NAMES[2:]:
            loss = loss + terms[k]
    return loss.mean(), terms


def test_train_results_are_lazy_and_follow_the_reference_formulas():
    raw, gt, _, _ = make_raw(37, 4, 0, "rgb", torch.device("cpu"))
    res = TrainResults(raw, {"weights_sum": raw.weights_sum})
    assert dict.__len__(res) == 1 and "image" in res and "depth" in res and "direct_rgb" in res and "nope" not in res
    assert res.get("nope", 5) == 5
    with pytest.raises(KeyError):
        res["nope"]
    img = raw.image_raw + (1 - raw.weights_sum).unsqueeze(-1) * raw.bg_color
    assert torch.equal(res["image"], img.view(1, 37, 3)) and dict.__contains__(res, "image")
    assert torch.equal(res["depth"], (torch.clamp(raw.depth_raw - raw.nears, min=0) / (raw.fars - raw.nears)).view(1, 37))
    assert torch.equal(res.get("direct_rgb"), (raw.all_map[..., 7:10] + (1 - raw.weights_sum).unsqueeze(-1) * raw.bg_color).view(1, 37, 3))
    res["image"].sum().backward()                       # the lazy entries carry autograd like the reference's
    assert raw.image_raw.grad is not None and raw.weights_sum.grad is not None
    nerf, _, _, _ = make_raw(5, 0, 0, "scalar", torch.device("cpu"), palette=False)
    assert "direct_rgb" not in TrainResults(nerf)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        train_loss(res, gt)
    with pytest.raises(RuntimeError, match="TrainResults"):
        train_loss({"image": img}, gt)


def test_train_loss_entry_points_validate_without_gpu():
    from palettenerf_amd import _lib
    lib = _lib.load()
    assert lib.pnr_train_loss_workspace_bytes(ctypes.c_uint32(4096)) == 64 + 16 * 8 * 4
    assert lib.pnr_train_loss_forward(None, None) == -1
    a = _lib.TrainLossArgs()
    a.N = 16
    assert lib.pnr_train_loss_forward(ctypes.byref(a), None) == -1           # null tensors
    p = ctypes.c_void_p(64)
    a.weights_sum = a.image_raw = a.gt_rgb = a.all_map = p
    a.num_basis, a.clip_dim, a.n_channel = 4, 8, 24                            # n_channel != 13 + clip + nb
    assert lib.pnr_train_loss_forward(ctypes.byref(a), None) == -1
    a.n_channel, a.bg_mode = 25, 3
    assert lib.pnr_train_loss_backward(ctypes.byref(a), None) == -1          # bg_mode
    a.bg_mode = 0
    assert lib.pnr_train_loss_forward(ctypes.byref(a), None) == -1           # no terms / workspace
    assert lib.pnr_train_loss_backward(ctypes.byref(a), None) == -1          # no gradient buffers
    a.N = 0
    assert lib.pnr_train_loss_backward(ctypes.byref(a), None) == 0


LAM = dict(sparsity=2e-4, offsets=0.03, view_dep=0.1, smooth=4e-3, weight=0.05, palette=1e-3)   # main_palette.py:83-89


def _compare(raw, gt, gt_clip, gt_w, bc, bco, scale=None):
    res = TrainResults(raw)
    l_ref, t_ref = torch_loss(res, gt, LAM, gt_clip, gt_w, bc, bco)
    leaves = [t for t in (raw.weights_sum, raw.image_raw, raw.all_map, bc) if t is not None]
    g_ref = torch.autograd.grad(l_ref * (scale if scale is not None else 1.0), leaves)
    res2 = TrainResults(raw)
    loss, info = train_loss(res2, gt, lambda_sparsity=LAM["sparsity"], lambda_offsets=LAM["offsets"], lambda_view_dep=LAM["view_dep"],
                            lambda_smooth=LAM["smooth"], lambda_weight=LAM["weight"], lambda_palette=LAM["palette"], gt_weights=gt_w, gt_clip=gt_clip,
                            basis_color=bc, basis_color_origin=bco)
    assert dict.__len__(res2) == 0                        # the fused loss never materialises the lazy entries
    g = torch.autograd.grad(loss * (scale if scale is not None else 1.0), leaves)
    assert abs(loss.item() - l_ref.item()) <= 2e-6 * abs(l_ref.item())
    assert info["terms"][0].item() == loss.item()
    for i, name in enumerate(TERM_NAMES[1:], 1):
        if name in t_ref:
            assert abs(float(info["terms"][i]) - float(t_ref[name].detach())) <= 2e-6 * abs(float(t_ref[name].detach())) + 1e-12, name
        else:
            assert float(info["terms"][i]) == 0.0
    for a, b in zip(g, g_ref):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12
    with torch.no_grad():
        assert float((info["image"] - res["image"]).abs().max()) == 0.0
        assert float((info["depth"] - res["depth"]).abs().max()) == 0.0
        assert torch.allclose(info["loss_ray"], ((res["image"] - gt.view(*raw.prefix, 3)) ** 2).mean(-1), rtol=3e-7, atol=1e-9)
        if raw.all_map is not None:
            assert float((info["direct_rgb"] - res["direct_rgb"]).abs().max()) == 0.0
    return loss.item()


@pytest.mark.gpu
@pytest.mark.parametrize("N,nb,clip,bg", [(4096, 4, 0, "scalar"), (4096, 6, 16, "rays"), (1000, 1, 8, "rgb"), (1, 10, 32, "rays"), (70001, 5, 0, "rgb")])
def test_train_loss_matches_the_torch_formulation(cuda, N, nb, clip, bg):
    raw, gt, gt_clip, gt_w = make_raw(N, nb, clip, bg, cuda, seed=N)
    bc = torch.rand(nb, 3, device=cuda, requires_grad=True)
    bco = torch.rand(nb, 3, device=cuda)
    first = _compare(raw, gt, gt_clip, gt_w, bc, bco)
    assert _compare(raw, gt, gt_clip, gt_w, bc, bco) == first            # reproducible: fixed reduction order, workspace reused
    _compare(raw, gt, None, None, None, None)                             # no guide, no clip target, no palette anchor
    _compare(raw, gt, gt_clip, gt_w, bc, bco, scale=torch.tensor(1024.0, device=cuda))   # a GradScaler-style scaled backward


@pytest.mark.gpu
def test_train_loss_nerf_model_and_prefix_shapes(cuda):
    raw, gt, _, _ = make_raw(2 * 2048, 0, 0, "rays", cuda, palette=False, prefix=(2, 2048))
    _compare(raw, gt, None, None, None, None)
    raw, gt, _, _ = make_raw(4096, 0, 0, "scalar", cuda, palette=False)
    res = TrainResults(raw)
    loss, info = train_loss(res, gt, want_outputs=False)
    assert info["image"] is None and info["depth"] is None and info["loss_ray"].shape == (1, 4096)
    assert abs(float(loss) - float(((res["image"] - gt[None]) ** 2).mean())) <= 2e-6 * float(loss)


@pytest.mark.gpu
def test_palette_training_step_with_the_fused_loss(cuda):
    """configs[3] shape end to end: the fused loss on run_cuda's TrainResults gives the parameter gradients of the torch formulation."""
    from palettenerf_amd import network, raymarching, renderer, scene
    torch.manual_seed(0)
    m = network.PaletteNetwork(renderer.default_opt(test=False), bound=2, cuda_ray=True, min_near=0.02)
    scene.seed_field_(m, 0)
    m = m.to(cuda).train()
    m.density_grid.copy_(torch.from_numpy(scene.slab_density_grid()).to(cuda))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)
    H, W = 756, 1008
    pose = np.eye(4, dtype=np.float32)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = [1, 0, 0], [0, -1, 0], [0, 0, -1], [0.3, 0.0, 1.5]
    ro, rd = scene.get_rays(torch.from_numpy(pose)[None], scene.intrinsics_from_fov(H, W, 0.9), H, W)
    inds = torch.randint(0, H * W, [4096])
    ro, rd = ro[:, inds].to(cuda), rd[:, inds].to(cuda)
    gt = torch.rand(1, 4096, 3, device=cuda)
    bco = (m.basis_color.detach() + 0.05).clone()

    def step(fused):
        for p in m.parameters():
            p.grad = None
        r = m.run_cuda(ro, rd, dt_gamma=1 / 128, perturb=False, force_all_rays=True, max_steps=1024, T_thresh=1e-4)
        assert isinstance(r, TrainResults) and r["omega_sparsity"].shape == (1, 4096) and r["basis_acc"].shape == (1, 4096, m.num_basis)
        if fused:
            loss, _ = train_loss(r, gt, lambda_sparsity=LAM["sparsity"], lambda_offsets=LAM["offsets"], lambda_view_dep=LAM["view_dep"],
                                 lambda_palette=LAM["palette"], basis_color=m.basis_color, basis_color_origin=bco)
        else:
            loss, _ = torch_loss(r, gt.reshape(-1, 3), dict(LAM, smooth=0.0, weight=0.0), bc=m.basis_color, bco=bco)
        loss.backward()
        return float(loss), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    l_ref, g_ref = step(False)
    l_fus, g_fus = step(True)
    assert abs(l_ref - l_fus) <= 2e-6 * abs(l_ref)
    assert set(g_ref) == set(g_fus) and "basis_color" in g_fus
    for name in g_ref:
        assert float((g_ref[name] - g_fus[name]).abs().max()) <= 2e-5 * float(g_ref[name].abs().max()) + 1e-12, name


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["palette", "nerf"])
def test_training_step_with_the_fused_loss_under_fp16_autocast(cuda, kind):
    """The reference's -O training mode (fp16 autocast + GradScaler, main_nerf.py:72-75): the fused loss takes the fp32 composites as they are, the scaled
    backward multiplies its gradients by the scaler's device scalar; three steps run, the loss is finite and the parameters move as with the torch loss."""
    import bench
    losses = {}
    for torch_loss in (False, True):
        torch.manual_seed(3)
        m, step = bench.make_training_step(kind, 4096, cuda, fp16=True, torch_loss=torch_loss)
        p0 = (m.color_net[0].weight if kind == "nerf" else m.diff_net[0].weight).detach().clone()
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        p1 = (m.color_net[0].weight if kind == "nerf" else m.diff_net[0].weight).detach()
        assert torch.isfinite(p1).all() and float((p1 - p0).abs().max()) > 0
        losses[torch_loss] = p1.clone()
    scale = float(losses[True].abs().max())
    assert float((losses[False] - losses[True]).abs().max()) <= 5e-2 * scale     # same trajectory to fp16 training noise (perturbed samples differ by the RNG stream only)
