#!/usr/bin/env python3
"""What the gradient comparisons of tests/test_gpu_frames.py (training steps against the reference-driven goldens) and
tests/test_gpu_fullsize.py (fused training path against torch modules + drop-in operators at configs[3] size) actually measure:
max |got - want| / max |want| per gradient.  The test tolerances are 4 x the largest figure printed here (profiles/r04_grad_tolerance.txt)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from palettenerf_amd import mlp, network, raymarching, renderer, scene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
cuda = torch.device("cuda:0")


def rel(got, want):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    return float(np.abs(got - want).max() / max(1e-30, np.abs(want).max()))


def put_scene(m):
    m.density_grid.copy_(torch.from_numpy(scene.brick_density_grid()).to(cuda))
    raymarching.packbits(m.density_grid, 0.5, m.density_bitfield)


def frame_rays(g):
    H, W = int(g["H"]), int(g["W"])
    ro, rd = scene.get_rays(torch.from_numpy(scene.lookat_pose())[None], scene.intrinsics_from_fov(H, W), H, W)
    return ro.to(cuda), rd.to(cuda)


worst = 0.0
for case in ("a", "b"):
    g = np.load(os.path.join(GOLDEN, f"train_nerf_{case}.npz"))
    m = network.NeRFNetwork(bound=2, cuda_ray=True, density_scale=float(g["density_scale"]), min_near=0.2)
    scene.seed_field_(m, int(g["seed"]))
    m = m.to(cuda).train()
    put_scene(m)
    ro, rd = frame_rays(g)
    r = m.run_cuda(ro, rd, dt_gamma=float(g["dt_gamma"]), perturb=False, force_all_rays=True, max_steps=1024, T_thresh=1e-4)
    loss = (r["image"] ** 2).mean() + 0.1 * r["weights_sum"].mean()
    loss.backward()
    rows = torch.from_numpy(g["grad_emb_rows"]).to(cuda)
    for name, got, want in (("grad_color0", m.color_net[0].weight.grad, g["grad_color0"]), ("grad_sigma1", m.sigma_net[1].weight.grad, g["grad_sigma1"]),
                            ("grad_emb", m.encoder.embeddings.grad[rows], g["grad_emb_vals"])):
        e = rel(got, want)
        worst = max(worst, e)
        print(f"train_nerf_{case:1s} {name:24s} rel err {e:.3e}   (max |g| {np.abs(want).max():.3e})")
    print(f"train_nerf_{case:1s} loss abs err {abs(float(loss) - float(g['loss'])):.3e}; grad_emb abs-sum ratio - 1 = {float(m.encoder.embeddings.grad.abs().sum()) / float(g['grad_emb_abs_sum']) - 1:.3e}")
    g = np.load(os.path.join(GOLDEN, f"train_palette_{case}.npz"))
    opt = renderer.default_opt(pred_clip=bool(g["pred_clip"]))
    m = network.PaletteNetwork(opt, bound=2, cuda_ray=True, density_scale=float(g["density_scale"]), min_near=0.2)
    scene.seed_field_(m, int(g["seed"]))
    m = m.to(cuda).train()
    put_scene(m)
    r = m.run_cuda(ro, rd, dt_gamma=float(g["dt_gamma"]), perturb=False, force_all_rays=True, max_steps=1024, T_thresh=1e-4)
    loss = (r["image"] ** 2).mean() + 0.01 * r["omega_sparsity"].mean() + 0.1 * r["offsets_norm"].mean() + (r["direct_rgb"] ** 2).mean() \
        + 0.1 * (r["clip_feat"] ** 2).mean() + 0.1 * r["basis_acc"].mean()
    loss.backward()
    rows = torch.from_numpy(g["grad_emb_rows"]).to(cuda)
    for name, got, want in (("grad_offsets_radiance", m.offsets_radiance_net.weight.grad, g["grad_offsets_radiance"]), ("grad_basis_color", m.basis_color.grad, g["grad_basis_color"]),
                            ("grad_diff0", m.diff_net[0].weight.grad, g["grad_diff0"]), ("grad_emb_palette", m.encoder_palette.embeddings.grad[rows], g["grad_emb_vals"])):
        e = rel(got, want)
        worst = max(worst, e)
        print(f"train_palette_{case:1s} {name:21s} rel err {e:.3e}   (max |g| {np.abs(want).max():.3e})")
    print(f"train_palette_{case:1s} loss abs err {abs(float(loss) - float(g['loss'])):.3e}")
print(f"== goldens: worst relative gradient error {worst:.3e}")

# configs[3] size: fused training path against torch modules + drop-in operators (tests/test_gpu_fullsize.py::test_palette_training_step_at_config3_size)
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
target = torch.rand(4096, 3, device=cuda)


def step(fused):
    mlp.enabled = fused
    m.fused_train_shade = m.fused_train_density = fused
    for p in m.parameters():
        p.grad = None
    r = m.run_cuda(ro, rd, dt_gamma=1 / 128, perturb=False, force_all_rays=True, max_steps=1024, T_thresh=1e-4)
    loss = ((r["image"][0] - target) ** 2).mean() + 1e-3 * r["omega_sparsity"].mean() + 1e-2 * r["offsets_norm"].mean() + ((r["direct_rgb"][0] - target) ** 2).mean()
    loss.backward()
    return float(loss), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}


l_ref, g_ref = step(False)
l_fus, g_fus = step(True)
l_fus2, g_fus2 = step(True)
mlp.enabled = True
worst3 = 0.0
for name in sorted(g_ref):
    scale = float(g_ref[name].abs().max())
    e = float((g_ref[name] - g_fus[name]).abs().max()) / max(scale, 1e-30)
    rr = float((g_fus2[name] - g_fus[name]).abs().max()) / max(scale, 1e-30)
    worst3 = max(worst3, e)
    print(f"config3 {name:34s} fused vs torch rel err {e:.3e}   fused run-to-run {rr:.3e}   (max |g| {scale:.3e})")
print(f"config3 loss: torch {l_ref:.8f} fused {l_fus:.8f} rel diff {abs(l_ref - l_fus) / abs(l_ref):.3e}")
print(f"== configs[3] size: worst relative gradient error {worst3:.3e}")
