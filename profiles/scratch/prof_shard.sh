R=$PWD; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/ph_s
rocprofv3 --kernel-trace --stats -d /tmp/ph_s -o p -- python3 $R/bench.py --workload garden --shard-emulation 8 --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd $R
python3 profiles/kernel_quantiles.py $(find /tmp/ph_s -name '*.db' | head -1) k_palette_field_fwd k_frame_grid_pair "k_frame_march<true, true, 2>" | cut -c1-200
