#!/bin/bash
# round 5, after the MLP latency work: full GPU suite, training-step kernel traces, the 5k-step training run, the training tolerances, the default bench line
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r05; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_final5.log 2>&1; echo "rc $?" >> $O/pytest_final5.log
cd /tmp
for m in palette nerf; do
  rm -rf /tmp/prof_train_step_$m
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train_step_$m -o p -- python3 $R/profiles/train_step_bench.py --model $m --steps 20 --warmup 5 > $O/train_step_$m.log 2>&1
  db=$(find /tmp/prof_train_step_$m -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 profiles/train_step_bench.py --model $m --steps 20 --warmup 5   (round 5)"; python3 $R/profiles/summarize.py $db; } > $O/train_step_$m.txt
done
cd $R
for m in palette nerf; do timeout 300 python profiles/train_step_bench.py --model $m --steps 50 --warmup 5 > $O/train_${m}_plain.log 2>&1; done
timeout 900 python profiles/train_palette.py > $O/train_palette_5k.txt 2>&1
timeout 600 python profiles/grad_tolerance.py > $O/grad_tolerance.txt 2>&1
timeout 1500 python bench.py > $O/bench_default.out 2> $O/bench_default.err; grep '^{"metric"' $O/bench_default.out > $O/bench_default.json
