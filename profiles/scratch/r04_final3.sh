#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r04; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_all.log 2>&1; echo "rc $?" >> $O/pytest_gpu_all.log
cd /tmp
prof() {  # name, program args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python3 "$@" > $R/gpurun_out/r04/$name.log 2>&1
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 $(echo "$@" | sed "s#$R/##g")   (round 4)"; python3 $R/profiles/summarize.py $db; } > $R/gpurun_out/r04/$name.txt
}
prof train_step_palette $R/profiles/train_step_bench.py --model palette --steps 20 --warmup 5
prof train_step_nerf $R/profiles/train_step_bench.py --model nerf --steps 20 --warmup 5
cd $R
timeout 300 python profiles/train_step_bench.py --model palette --steps 50 --warmup 5 > $O/train_palette_h.log 2>&1
timeout 300 python profiles/train_step_bench.py --model nerf --steps 50 --warmup 5 > $O/train_nerf_h.log 2>&1
timeout 300 python profiles/grad_tolerance.py > $O/grad_tolerance2.log 2>&1
