run() { python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']/1e9,3), round(d['step_ms']['median'],3), round(d['roofline']['frac'],3))"; }
python -m pytest tests/test_gpu_frames.py tests/test_gpu_ops.py -x -q 2>&1 | tail -2
echo "lego: $(run) | $(run)"
echo "garden: $(run --workload garden --steps 20) | $(run --workload garden --steps 20)"
echo "palette: $(run --workload lego_palette --steps 20) | $(run --workload lego_palette --steps 20)"
WL=lego bash profiles/scratch/prof_quick.sh 2>&1 | grep -A5 "== hosted"
WL=garden bash profiles/scratch/prof_quick.sh 2>&1 | grep -A5 "== hosted"
