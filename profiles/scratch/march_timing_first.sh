touch palettenerf_amd/csrc/frame.hip; PNR_EXTRA_HIPCC_FLAGS="-DPNR_MARCH_TIMING" python -m palettenerf_amd.build 2>&1 | grep -E " error" | head -3
for pose in 5 11; do python profiles/march_timing.py --tile8 --pose $pose 0 2>&1 | grep -v amdgpu.ids | head -36; done
