# A/B of the hosted march tail (scratch script for gpurun; one line per variant: ms/step, G samples/s, median, roofline frac)
run() { python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']/1e9,3), round(d['step_ms']['median'],3), round(d['roofline']['frac'],3))"; }
python -m pytest tests/test_gpu_frames.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
echo "lego hosted b2: $(run) | $(run)"
echo "lego no hosted: $(PNR_NO_HOSTED_TAIL=1 run) | $(PNR_NO_HOSTED_TAIL=1 run)"
for b in 1 3 4; do echo "lego hosted b$b: $(PNR_MARCH_BUDGET=$b run)"; done
echo "garden hosted b2: $(run --workload garden --steps 20)"
echo "garden no hosted: $(PNR_NO_HOSTED_TAIL=1 run --workload garden --steps 20)"
for b in 4 8; do echo "garden hosted b$b: $(PNR_MARCH_BUDGET=$b run --workload garden --steps 20)"; done
echo "palette hosted b2: $(run --workload lego_palette --steps 20)"
echo "palette no hosted: $(PNR_NO_HOSTED_TAIL=1 run --workload lego_palette --steps 20)"
WL=lego bash profiles/scratch/prof_quick.sh
