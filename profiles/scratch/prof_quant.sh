R=$PWD; export TMPDIR=/tmp
for v in hosted nohosted; do
  cd /tmp; rm -rf /tmp/ph_$v
  if [ $v = nohosted ]; then export PNR_NO_HOSTED_TAIL=1; else unset PNR_NO_HOSTED_TAIL; fi
  rocprofv3 --kernel-trace --stats -d /tmp/ph_$v -o p -- python3 $R/bench.py --workload ${WL:-lego} --steps 15 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  cd $R
  echo "== $v"; python3 profiles/kernel_quantiles.py $(find /tmp/ph_$v -name '*.db' | head -1) k_frame_grid k_frame_march k_frame_field | cut -c1-400
done
