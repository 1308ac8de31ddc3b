#!/usr/bin/env python3
"""ms per update_extra_state, 2 x 128^3 cells: the device-resident sweep of csrc/occupancy.hip -- full (iter_density < 16) and
partial -- against the same sweep with the field evaluated through density() (torch sigma_net between the point and scatter kernels: what a
field without a fused kernel costs).  Wall time over 20 back-to-back calls and HIP-event time of one; host time = what the call costs the
enqueueing thread (nothing waits for the device)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from palettenerf_amd import network, scene  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    m = network.NeRFNetwork(bound=2, cuda_ray=True, min_near=0.2)
    scene.seed_field_(m, 0)
    m = m.to(dev).train()
    for generic, mode, start in ((False, "full", 0), (False, "partial", 16), (True, "full", 0), (True, "partial", 16)):
        m.occupancy_generic = generic
        for rep in range(3):
            m.iter_density = start
            m.update_extra_state()
        torch.cuda.synchronize()
        n = 20
        host = 0.0
        t0 = time.perf_counter()
        for rep in range(n):
            m.iter_density = start
            h0 = time.perf_counter()
            m.update_extra_state()
            host += time.perf_counter() - h0
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        m.iter_density = start
        noise = torch.rand(2, 128 ** 3, 3, device=dev) if start == 0 else None
        e0.record()
        m.update_extra_state(noise=noise)
        e1.record()
        torch.cuda.synchronize()
        print(f"update_extra_state {'generic density()' if generic else 'fused sweep'} {mode}: wall {wall:.3f} ms/call, host {host / n * 1e3:.3f} ms/call, "
              f"device {e0.elapsed_time(e1):.3f} ms (noise drawn outside: full only), occupied {int((m.density_grid > 0).sum())}, mean {m.mean_density:.4f}")


if __name__ == "__main__":
    main()
