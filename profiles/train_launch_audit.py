#!/usr/bin/env python3
"""Which launches make up one configs[3] training step?  Runs bench.make_training_step under torch.profiler (CPU + device activities, stacks
on) and prints, per launching aten op / Python line, the number of device kernels per step -- the list the launch diet works from."""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["palette", "nerf"], default="palette")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--torch-loss", action="store_true")
    args = ap.parse_args()
    from torch.profiler import ProfilerActivity, profile
    dev = torch.device("cuda:0")
    m, step = bench.make_training_step(args.model, 4096, dev, torch_loss=args.torch_loss)
    for i in range(6):
        step(i)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for i in range(args.steps):
            step(6 + i)
        torch.cuda.synchronize()
    rows = collections.Counter()
    times = collections.Counter()
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for e in prof.events():
        if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
            continue
        if any(c.kernels for c in (e.cpu_children or [])):
            continue   # count at the innermost launching op
        top = e
        while top.cpu_parent is not None:
            top = top.cpu_parent
        where = ""
        q = e
        while q is not None and not q.stack:
            q = q.cpu_parent
        for fr in (q.stack if q is not None else []):
            if (here in fr or "palettenerf_amd" in fr or "bench.py" in fr) and "profiles/" not in fr:
                where = fr.replace(here + "/", "")
                break
        names = ",".join(sorted({k.name.split("<")[0].split("(")[0][-48:] for k in e.kernels}))
        key = (top.name[:44], e.name[:40], where[:70], names[:60])
        rows[key] += len(e.kernels)
        times[key] += sum(k.duration for k in e.kernels)
    total = sum(rows.values())
    print(f"# {args.model}: {total / args.steps:.1f} launches per step over {args.steps} steps (memcpy/memset included)")
    print(f"{'n/step':>7} {'us/step':>8}  top-level op | launching op | first repo frame | kernel")
    for key, n in sorted(rows.items(), key=lambda kv: -kv[1]):
        print(f"{n / args.steps:7.2f} {times[key] / args.steps:8.1f}  " + " | ".join(key))


if __name__ == "__main__":
    main()
