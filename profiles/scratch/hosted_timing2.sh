touch palettenerf_amd/csrc/frame.hip; PNR_EXTRA_HIPCC_FLAGS="-DPNR_HOSTED_TIMING $EXTRA" python -m palettenerf_amd.build 2>&1 | grep -E " error" | head -3
echo "== hosted"; HOSTED_TIMING_BRIEF=1 python profiles/hosted_timing.py $(seq 1 2 27) 2>&1 | grep iteration | cut -c1-250
python profiles/hosted_timing.py 5 15 2>&1 | grep -v "^iteration" | tail -12
echo "== no hosted"; PNR_NO_HOSTED_TAIL=1 HOSTED_TIMING_BRIEF=1 python profiles/hosted_timing.py $(seq 1 2 27) 2>&1 | grep iteration | cut -c1-250
