#!/usr/bin/env python3
"""The iteration schedule of one 800x800 S0 frame (n_alive, n_step per march/shade/compact round of nerf/renderer.py:354-380): how much of the
frame runs with few rays alive.  Uses the host-driven 'device' loop, which knows n_alive of every iteration."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from palettenerf_amd import raymarching, scene  # noqa: E402


def main():
    sys.argv = ["bench.py"]
    args = bench.parse()
    dev = torch.device("cuda:0")
    m = bench.build_model(args, dev)
    m.march_mode, m.fused_field = "device", True
    log = []
    orig = raymarching.march_rays

    def spy(n_alive, n_step, *a, **k):
        log.append((n_alive, n_step))
        return orig(n_alive, n_step, *a, **k)

    from palettenerf_amd import renderer
    renderer.raymarching.march_rays = spy
    pose = torch.from_numpy(scene.lookat_pose())[None]
    ro, rd = scene.get_rays(pose, scene.intrinsics_from_fov(800, 800), 800, 800)
    with torch.no_grad():
        m.render(ro.to(dev), rd.to(dev), perturb=False, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    rows = 0
    for i, (na, ns) in enumerate(log):
        rows += na * ns
        print(f"iter {i:2d}: n_alive {na:7d}  n_step {ns}  rows {na * ns:7d}  cumulative rows {rows}")


if __name__ == "__main__":
    main()
