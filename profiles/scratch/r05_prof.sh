#!/bin/bash
# rocprofv3 kernel-trace summaries of the three workloads (round 5)
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O; TAG=${TAG:-a}
cd /tmp && export TMPDIR=/tmp
for wl in ${WLS:-garden lego_palette lego}; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_$wl -o $wl -- python3 $R/bench.py --workload $wl --steps 10 --warmup 5 --no-cpu-baseline --no-extras --no-traffic > $O/prof_${TAG}_$wl.log 2>&1
  f=$(find $O/prof_${TAG}_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" > $O/prof_${TAG}_$wl.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':90s} {'calls':>7s} {'avg_us':>9s} {'total_ms':>9s} {'pct':>6s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:9.2f} {float(r['TotalDurationNs'])/1e6:9.2f} {100*float(r['TotalDurationNs'])/tot:6.1f}")
PY
done
