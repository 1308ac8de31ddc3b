PNR_EXTRA_HIPCC_FLAGS="-DPNR_MARCH_TIMING -DPNR_MARCH_WAVES=4" python -m palettenerf_amd.build --force >/dev/null 2>&1
export PNR_NO_COOP_MARCH=1
python profiles/march_timing.py --tile8 --pose 5 3 12 2>&1 | grep -E "iteration|slowest|the 200|max probes" | head -40
