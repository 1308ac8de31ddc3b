run() { python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']/1e9,3), round(d['step_ms']['median'],3), round(d['roofline']['frac'],3))"; }
python -m pytest tests/test_gpu_frames.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
for w in 6 5 4; do
  touch palettenerf_amd/csrc/frame.hip; PNR_EXTRA_HIPCC_FLAGS="-DPNR_MARCH_WAVES_Q=$w" python -m palettenerf_amd.build >/dev/null 2>&1
  echo "queue waves $w: lego $(run) | $(run) ; garden $(run --workload garden --steps 20) ; palette $(run --workload lego_palette --steps 20)"
done
echo "nocoop waves 4: lego $(PNR_NO_COOP_MARCH=1 run)"
