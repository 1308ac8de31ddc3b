touch palettenerf_amd/csrc/frame.hip; PNR_EXTRA_HIPCC_FLAGS="-DPNR_MARCH_TIMING" python -m palettenerf_amd.build 2>&1 | grep -E " error" | head -3
python profiles/march_timing.py --tile8 --pose 5 3 2>&1 | head -42
