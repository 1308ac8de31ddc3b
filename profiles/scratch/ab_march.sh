# A/B of the wave-cooperative march tail (scratch script for gpurun; prints one line per variant)
run() { python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], round(d['value']/1e9,3), d.get('step_ms'))"; }
python -m pytest tests/test_gpu_frames.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -15
echo COOP6 $(run); echo NOCOOP6 $(PNR_NO_COOP_MARCH=1 run)
echo COOP6 $(run); echo NOCOOP6 $(PNR_NO_COOP_MARCH=1 run)
echo GARDEN_COOP6 $(run --workload garden --steps 20); echo GARDEN_NOCOOP6 $(PNR_NO_COOP_MARCH=1 run --workload garden --steps 20)
for w in 5 4; do
  touch palettenerf_amd/csrc/frame.hip; PNR_EXTRA_HIPCC_FLAGS="-DPNR_MARCH_WAVES=$w" python -m palettenerf_amd.build >/dev/null 2>&1
  echo COOP$w $(run); echo NOCOOP$w $(PNR_NO_COOP_MARCH=1 run); echo COOP$w $(run)
  echo GARDEN_COOP$w $(run --workload garden --steps 20)
done
