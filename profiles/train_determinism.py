"""Are a training step's gradients reproducible bit for bit from run to run (same batch, perturb off)?  Prints per parameter."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palettenerf_amd import network, raymarching, renderer, scene
from palettenerf_amd.train_loss import train_loss
cuda = torch.device("cuda:0")
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
p = torch.randint(0, H * W, [4096])
ro, rd = ro[:, p].to(cuda).contiguous(), rd[:, p].to(cuda).contiguous()
gt = torch.rand(1, 4096, 3, device=cuda)
def step():
    for q in m.parameters(): q.grad = None
    r = m.run_cuda(ro, rd, dt_gamma=1/128, perturb=False, force_all_rays=True, max_steps=1024, T_thresh=1e-4)
    loss, _ = train_loss(r, gt, lambda_sparsity=2e-4, lambda_offsets=0.03)
    loss.backward()
    return loss.item(), {k: q.grad.clone() for k, q in m.named_parameters() if q.grad is not None}
a = step(); b = step(); c = step()
print("loss equal:", a[0] == b[0] == c[0])
for k in a[1]:
    e1 = torch.equal(a[1][k], b[1][k]); e2 = torch.equal(b[1][k], c[1][k])
    d = float((a[1][k] - b[1][k]).abs().max()) / (float(a[1][k].abs().max()) + 1e-30)
    print(f"{k:40s} equal {e1} {e2}  rel diff {d:.2e}")
