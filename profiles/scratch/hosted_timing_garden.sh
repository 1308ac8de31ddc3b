touch palettenerf_amd/csrc/frame.hip; PNR_EXTRA_HIPCC_FLAGS="-DPNR_HOSTED_TIMING" python -m palettenerf_amd.build 2>&1 | grep -E " error" | head -3
HOSTED_TIMING_BRIEF=1 python profiles/hosted_timing.py --workload garden --pose 5 $(seq 1 4 29) 2>&1 | grep iteration | cut -c1-260
python profiles/hosted_timing.py --workload garden --pose 5 9 2>&1 | grep -v amdgpu | tail -6
