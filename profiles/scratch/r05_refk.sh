#!/bin/bash
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp; rm -rf /tmp/prof_refk
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_refk -o p -- python3 $R/profiles/reference_kernels.py --json > $O/refk_prof.log 2>&1
db=$(find /tmp/prof_refk -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 profiles/reference_kernels.py   (round 5, at HEAD: the reference's kernels -- names without pnr:: -- next to this repository's, same inputs)"; python3 $R/profiles/summarize.py $db; } > $O/reference_kernels_rocprof.txt
