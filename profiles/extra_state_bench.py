#!/usr/bin/env python3
"""ms per update_extra_state: full sweep (iter_density < 16) and partial update, NeRF and PaletteNeRF."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from palettenerf_amd import network, renderer, scene  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for kind in ("nerf",):  # PaletteRenderer has no update_extra_state (its density grid is frozen from the NeRF stage)
        if kind == "palette":
            m = network.PaletteNetwork(renderer.default_opt(test=False), bound=2, cuda_ray=True, min_near=0.2)
        else:
            m = network.NeRFNetwork(bound=2, cuda_ray=True, min_near=0.2)
        scene.seed_field_(m, 0)
        m = m.to(dev).train()
        for fused, mode, start in ((False, "full", 0), (False, "partial", 16), (True, "full", 0), (True, "partial", 16)):
            m.fused_field = fused
            for rep in range(3):
                m.iter_density = start
                m.update_extra_state()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for rep in range(n):
                m.iter_density = start
                m.update_extra_state()
            torch.cuda.synchronize()
            print(f"{kind} update_extra_state fused_density={fused} {mode}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms, occupied {int((m.density_grid > 0).sum())}")


if __name__ == "__main__":
    main()
