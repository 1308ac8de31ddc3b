#!/bin/bash
# One eighth of the garden frame (shard 0 of 8) and the whole frame under rocprofv3 --kernel-trace: per-launch kernel times, in-frame and between-frame gaps.
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp
for s in 8 1; do
  rm -rf /tmp/prof_shard$s
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_shard$s -o p -- python3 $R/profiles/shard_profile.py $s > $O/shard$s.log 2>&1
  db=$(find /tmp/prof_shard$s -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 profiles/shard_profile.py $s   (round 5)"; grep "^shards" $O/shard$s.log; python3 $R/profiles/summarize.py $db | head -16; python3 $R/profiles/frame_gaps.py $db; python3 $R/profiles/frame_boundary.py $db; } > $O/shard$s.txt
done
