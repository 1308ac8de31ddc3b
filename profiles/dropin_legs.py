"""The four drop-in legs of bench.py alone (configs[1] / [2] through the operator API, with and without dropin.fuse_field)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
args = bench.parse(["--no-cpu-baseline"])
device = torch.device("cuda")
from palettenerf_amd import dist as pdist
idx, _ = pdist.shard_indices(args.wl["H"], args.wl["W"], 0, 1)
bank = bench.RayBank(args, 1, idx, device)
out = {}
for kind, ff in (("palette", True), ("palette", False), ("nerf", True)):
    r = bench.dropin_leg(args, device, bank, kind, 8, fuse_field=ff)
    out[kind + ("_fuse_field" if ff else "")] = {k: v for k, v in r.items() if k not in ("what", "lookup_op")}
print(json.dumps(out, indent=1))
