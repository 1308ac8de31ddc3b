#!/usr/bin/env python3
"""BASELINE configs[0] on the GPU: NeRFRenderer.run (no occupancy grid; 400x400, --num_steps 512 --upsample_steps 0 and the 128+128 defaults),
staged in max_ray_batch = 4096 pieces as the reference's render() does, over the HIP near/far, hash-grid and SH operators + torch MLPs."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from palettenerf_amd import network, scene  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    m = network.NeRFNetwork(bound=2, cuda_ray=False, density_scale=100.0, min_near=0.2)
    scene.seed_field_(m, 0)
    m = m.to(dev).eval()
    pose = torch.from_numpy(scene.lookat_pose())[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(400, 400), 400, 400)
    ro, rd = ro.to(dev), rd.to(dev)
    for fused, ns, us, batch in ((False, 512, 0, 4096), (False, 128, 128, 4096), (True, 512, 0, 4096), (True, 128, 128, 4096), (True, 512, 0, 160000)):
        m.fused_field = fused
        for rep in range(2):
            with torch.no_grad():
                m.render(ro, rd, staged=True, max_ray_batch=batch, num_steps=ns, upsample_steps=us, perturb=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for rep in range(n):
            with torch.no_grad():
                m.render(ro, rd, staged=True, max_ray_batch=batch, num_steps=ns, upsample_steps=us, perturb=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"uniform path 400x400 fused_field={fused} num_steps={ns} upsample_steps={us} max_ray_batch={batch}: {dt * 1e3:.1f} ms/frame, "
              f"{160000 * (ns + us) / dt / 1e6:.0f} M evaluated samples/s")


if __name__ == "__main__":
    main()
