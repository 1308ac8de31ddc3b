#!/usr/bin/env python3
"""Throughput with several frames in flight (two host threads, each with its own model copy, workspace and stream, rendering alternate poses of the
bench's camera path) against the plain one-frame-at-a-time loop.  usage: frames_in_flight.py [--workload lego] [--steps 40]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from palettenerf_amd import dist as pdist  # noqa: E402
from palettenerf_amd.fused import tile_ray_order  # noqa: E402

argv = [a for a in sys.argv[1:]]
args = bench.parse(argv + ["--no-cpu-baseline"])
dev = torch.device("cuda", 0)
wl = args.wl
H, W = wl["H"], wl["W"]
idx, _ = pdist.shard_indices(H, W, 0, 1)
bank = bench.RayBank(args, 1, idx, dev)
kw = dict(perturb=False, dt_gamma=wl["dt_gamma"], max_steps=1024, T_thresh=1e-4)
if args.model == "palette":
    kw["gui_mode"] = False
steps = args.steps
for i in range(steps + 4):
    bank.get(i)


def make():
    m = bench.build_model(args, dev)
    m._fused.ray_order = tile_ray_order(idx, W, 8).to(dev)
    return m


def run(m, frames, stream, out, lat=None):
    n = 0
    with torch.cuda.stream(stream), torch.no_grad():
        for i in frames:
            ro, rd = bank.get(i)
            t = time.perf_counter()
            r = m.render(ro, rd, **kw)
            n += int(r["rendered"].sum())
            if lat is not None:
                lat.append(time.perf_counter() - t)
    out.append(n)


F = int(os.environ.get("FRAMES_IN_FLIGHT", "2"))
if os.environ.get("SEPARATE_WEIGHTS"):   # A/B: every handle with its own copy of the weights instead of clone_for_concurrent_frames
    models = [make() for _ in range(F)]
else:
    from palettenerf_amd.pipeline import clone_for_concurrent_frames
    models = [make()]
    models += [clone_for_concurrent_frames(models[0]) for _ in range(F - 1)]
streams = [torch.cuda.Stream() for _ in range(F)]
for m, s in zip(models, streams):
    run(m, range(4), s, [])
torch.cuda.synchronize()
for rnd in range(3):
    out = []
    t0 = time.perf_counter()
    run(models[0], range(4, 4 + steps), streams[0], out)
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    out2 = []
    t0 = time.perf_counter()
    lat = []
    th = [threading.Thread(target=run, args=(models[k], range(4 + k, 4 + steps, F), streams[k], out2, lat)) for k in range(F)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    print(f"{args.workload}: one frame at a time {t1 / steps * 1e3:.3f} ms/frame ({sum(out) / t1 / 1e9:.3f} G samples/s); {F} in flight {t2 / steps * 1e3:.3f} ms/frame ({sum(out2) / t2 / 1e9:.3f} G samples/s), latency of a frame {sorted(lat)[len(lat) // 2] * 1e3:.2f} ms median")
