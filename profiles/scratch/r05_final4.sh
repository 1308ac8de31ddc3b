#!/bin/bash
# round 5, after the frame-turnaround work: kernel-trace summaries, the shard profile, the default bench line (as the driver runs it) and the one-rank RCCL default
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
bash profiles/r05_profile.sh > $O/profile.log 2>&1
bash profiles/r05_shard.sh > $O/shard.log 2>&1
timeout 1500 python bench.py > $O/bench_default.out 2> $O/bench_default.err; grep '^{"metric"' $O/bench_default.out > $O/bench_default.json
PNR_BENCH_FORCE_DIST=1 timeout 900 python bench.py --dist-default --gpus 1 > $O/bench_dist.out 2> $O/bench_dist.err; grep '^{"metric"' $O/bench_dist.out > $O/bench_dist_default.json
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log
