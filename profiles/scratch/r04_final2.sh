#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
S=$(date +%s); python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "rc $? seconds $(( $(date +%s) - S ))" >> $O/bench_default.err
cd /tmp
prof() {  # name, program args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python3 "$@" > $R/gpurun_out/r04/$name.log 2>&1
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 $(echo "$@" | sed "s#$R/##g")   (round 4)"; python3 $R/profiles/summarize.py $db; } > $R/gpurun_out/r04/$name.txt
}
prof bench_lego $R/bench.py --workload lego --steps 15 --warmup 3 --no-cpu-baseline --no-extras --no-traffic
