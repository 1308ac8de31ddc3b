#!/bin/bash
# rocprofv3 kernel summaries of the two configs[3] training steps -> gpurun_out/r03/train_step_{palette,nerf}.txt (+ the torch-loss variant of palette)
R=$PWD; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r03
cd /tmp
prof() {
  name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python3 "$@" > $R/gpurun_out/r03/$name.log 2>&1
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 $(echo "$@" | sed "s#$R/##g")   (round 3)"; python3 $R/profiles/summarize.py $db; } > $R/gpurun_out/r03/$name.txt
}
prof train_step_palette $R/profiles/train_step_bench.py --model palette --steps 20 --warmup 5
prof train_step_nerf $R/profiles/train_step_bench.py --model nerf --steps 20 --warmup 5
prof train_step_palette_torch_loss $R/profiles/train_step_bench.py --model palette --steps 20 --warmup 5 --torch-loss
cd $R
for m in palette nerf; do python3 profiles/train_launch_audit.py --model $m 2>&1 | grep -v "^/\|Warn\|_warn\|^\[W" > gpurun_out/r03/train_launch_audit_$m.txt; done
python3 profiles/train_launch_audit.py --model palette --torch-loss 2>&1 | grep -v "^/\|Warn\|_warn\|^\[W" > gpurun_out/r03/train_launch_audit_palette_torch_loss.txt
tail -1 gpurun_out/r03/train_step_*.log
