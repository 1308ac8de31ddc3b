#!/usr/bin/env python3
"""Host time of one configs[3] training step: the GPU is idle when the step starts (synchronize in front), the clock stops when step() returns
(everything queued; the only wait inside is march_rays_train's look at the sample counter).  Beside it: the same step's wall time in a free-running
loop and its kernel time.  If host time ~ kernel time, the step is paced by Python as much as by the GPU."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for kind in ("palette", "nerf"):
        m, step = bench.make_training_step(kind, 4096, dev)
        for i in range(8):
            step(i)
        host = []
        for i in range(8, 40):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step(i)
            host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40, 140):
            step(i)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 100
        host.sort()
        print(f"{kind}: host time per step (GPU idle at its start) median {host[len(host) // 2] * 1e3:.2f} ms, min {host[0] * 1e3:.2f} ms; free-running wall {wall * 1e3:.2f} ms/step")


if __name__ == "__main__":
    main()
