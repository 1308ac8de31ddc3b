"""Where the time of a drop-in frame with dropin.fuse_field goes (the reference's loop, the network's forward() on the fused field kernel): wall per frame,
GPU-busy time and launches per frame, and the host side by function (cProfile, top of tottime).  usage: dropin_host_profile.py [nerf|palette]"""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

kind = sys.argv[1] if len(sys.argv) > 1 else "nerf"
args = bench.parse(["--mode", "compat", "--no-cpu-baseline"] + (["--model", "palette"] if kind == "palette" else []))
dev = torch.device("cuda", 0)
m = bench.build_model(args, dev)
from palettenerf_amd import dist as pdist, dropin
dropin.fuse_field(m, "f16x3")
idx, _ = pdist.shard_indices(800, 800, 0, 1)
bank = bench.RayBank(args, 1, idx, dev)
kw = dict(perturb=False, dt_gamma=0.0, max_steps=1024, T_thresh=1e-4)
if kind == "palette":
    kw["gui_mode"] = False
ro, rd = bank.get(0)
with torch.no_grad():
    for _ in range(3):
        m.render(ro, rd, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        m.render(ro, rd, **kw)
    torch.cuda.synchronize()
    print(f"wall {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per frame")
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as tp:
        m.render(ro, rd, **kw)
        torch.cuda.synchronize()
    ks = [e for e in tp.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    print(f"kernels {len(ks)} per frame, GPU busy {sum(e.device_time for e in ks) / 1e3:.2f} ms")
    import collections
    c = collections.Counter()
    for e in ks:
        c[e.name[:70]] += e.device_time
    for n, t in c.most_common(14):
        print(f"   {t / 1e3:7.3f} ms  {n}")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        m.render(ro, rd, **kw)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
