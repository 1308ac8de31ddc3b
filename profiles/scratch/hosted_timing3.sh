run() { python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']/1e9,3), round(d['step_ms']['median'],3), round(d['roofline']['frac'],3))"; }
for flags in "" "-DPNR_HOSTED_JUMPS=false"; do
touch palettenerf_amd/csrc/frame.hip; PNR_EXTRA_HIPCC_FLAGS="$flags" python -m palettenerf_amd.build 2>&1 | grep -E " error" | head -3
echo "== [$flags] lego: $(run) | $(run)  garden: $(run --workload garden --steps 20) palette: $(run --workload lego_palette --steps 20)"
touch palettenerf_amd/csrc/frame.hip; PNR_EXTRA_HIPCC_FLAGS="-DPNR_HOSTED_TIMING $flags" python -m palettenerf_amd.build 2>&1 | grep -E " error" | head -3
for pose in 10 16; do
echo "== [$flags] pose $pose"; HOSTED_TIMING_BRIEF=1 python profiles/hosted_timing.py --pose $pose $(seq 1 6 27) 2>&1 | grep iteration | cut -c1-250
done
done
python -m pytest tests/test_gpu_frames.py -x -q -k "hosted" 2>&1 | tail -2
