#!/bin/bash
# usage: profiles/pmc_pass.sh <out-subdir under gpurun_out> "<counters of pass 1>" "<counters of pass 2>" ... -- <bench.py args>
# One rocprofv3 --pmc run per counter group (kernel-trace only, csv), then profiles/pmc_summary.py over all passes.
R=$PWD; export TMPDIR=/tmp
out=$1; shift
groups=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do groups+=("$1"); shift; done
shift
cd /tmp
i=0
for g in "${groups[@]}"; do
  first=${g%% *}
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $R/gpurun_out/$out/$first -o p -- python3 $R/${PNR_PMC_SCRIPT:-bench.py} "$@" > $R/gpurun_out/$out.$i.log 2>&1
  i=$((i+1))
done
python3 $R/profiles/pmc_summary.py $R/gpurun_out/$out > $R/gpurun_out/$out.txt
