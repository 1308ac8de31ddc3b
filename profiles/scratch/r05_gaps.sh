#!/bin/bash
# between-frame gaps of the native loop on the three bench workloads + a host profile of the palette shard frame
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r05; mkdir -p $O; TAG=${TAG:-a}
cd /tmp
for wl in lego lego_palette garden; do
  rm -rf /tmp/prof_g$wl
  timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_g$wl -o p -- python3 $R/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-traffic > $O/gaps_$wl.log 2>&1
  db=$(find /tmp/prof_g$wl -name '*.db' | head -1)
  { echo "## $wl"; grep -o '"ms_per_step": [0-9.]*' $O/gaps_$wl.log | head -1; python3 $R/profiles/frame_gaps.py $db; python3 - $db <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
ends = [i for i, r in enumerate(rows) if "k_frame_unsort" in r[0]]
e = ends[len(ends) // 2]
t0 = rows[e][2]
for r in rows[e - 2:e + 24]:
    print(f"  {(r[1] - t0) / 1e3:9.1f} us  +{(r[2] - r[1]) / 1e3:7.1f}  {r[0][:90]}")
PY
  } > $O/gaps_${wl}_$TAG.txt 2>&1
done
cd $R
python3 - > $O/host_shard_$TAG.txt 2>&1 <<'PY'
import cProfile, pstats, sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from palettenerf_amd import dist as pdist
from palettenerf_amd.fused import tile_ray_order
args = bench.parse(["--workload", "garden", "--no-cpu-baseline", "--static-pose"])
dev = torch.device("cuda", 0)
m = bench.build_model(args, dev)
H, W = args.wl["H"], args.wl["W"]
idx, _ = pdist.shard_indices(H, W, 0, 8)
bank = bench.RayBank(args, 1, idx, dev)
m._fused.ray_order = tile_ray_order(idx, W, 8).to(dev)
kw = dict(perturb=False, dt_gamma=args.wl["dt_gamma"], max_steps=1024, T_thresh=1e-4, gui_mode=False)
with torch.no_grad():
    for i in range(5):
        m.render(*bank.get(i), **kw)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for i in range(50):
        m.render(*bank.get(i), **kw)
    torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
PY
