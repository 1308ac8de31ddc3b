R=$PWD; export TMPDIR=/tmp
prof() {  # $1 = tag, rest = env assignments
  tag=$1; shift
  cd /tmp; rm -rf /tmp/ph_x
  ( export "$@"; rocprofv3 --kernel-trace --stats -d /tmp/ph_x -o p -- python3 $R/bench.py --workload ${WL:-lego} --steps 15 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1 )
  cd $R
  echo "== $tag"; python3 profiles/summarize.py $(find /tmp/ph_x -name '*.db' | head -1) | head -8 | tail -5 | cut -c1-110
}
prof "nohosted" PNR_NO_HOSTED_TAIL=1
prof "hosted" PNR_MARCH_BUDGET=2
