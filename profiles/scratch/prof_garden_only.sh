R=$PWD; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r03
cd /tmp; rm -rf /tmp/prof_bench_garden
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench_garden -o p -- python3 $R/bench.py --workload garden --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload garden --steps 10 --warmup 3 --no-cpu-baseline --no-extras   (round 3)"; python3 $R/profiles/summarize.py $(find /tmp/prof_bench_garden -name '*.db' | head -1); } > $R/gpurun_out/r03/bench_garden.txt
head -9 $R/gpurun_out/r03/bench_garden.txt | cut -c1-120
