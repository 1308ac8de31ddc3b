"""One eighth of the garden frame (configs[4] split over 8 ranks: shard 0's interleaved 32x32 tiles) rendered N times: what a strong-scaling
shard's kernels cost per launch, next to the full frame's.  Run under rocprofv3 --kernel-trace --stats (profiles/r04_shard.sh)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from palettenerf_amd import dist as pdist
from palettenerf_amd.fused import tile_ray_order

shards = int(sys.argv[1]) if len(sys.argv) > 1 else 8
args = bench.parse(["--workload", "garden", "--no-cpu-baseline", "--static-pose"])
dev = torch.device("cuda", 0)
m = bench.build_model(args, dev)
H, W = args.wl["H"], args.wl["W"]
idx, _ = pdist.shard_indices(H, W, 0, shards)
bank = bench.RayBank(args, 1, idx, dev)
m._fused.ray_order = tile_ray_order(idx, W, 8).to(dev)
kw = dict(perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4, gui_mode=False)
bench.timed_frames(m, bank, kw, 3, False)
if "--one-call" in sys.argv:     # rounds 1-5: m.render() per frame, the host turns around between two frames
    med, mean, rend = bench.timed_frames_median(m, bank, kw, 15)
    how = "one call per frame"
else:                            # round 6: render_prepare / render_launch / render_wait / render_result (pipeline.render_queue): frame i + 1 prepared under frame i's kernels
    med, mean, rend = bench.timed_frames_queue(m, bank, kw, 15)
    how = "prepare / launch / finish queue"
print(f"shards {shards}: {idx.numel()} rays, median {med:.3f} ms, mean {mean:.3f} ms, {rend} samples per frame ({how})")
