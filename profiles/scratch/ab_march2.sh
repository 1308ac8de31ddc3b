run() { python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']/1e9,3), round(d['step_ms']['median'],3))"; }
for w in 4 5; do for r in 1 2 4; do
  touch palettenerf_amd/csrc/frame.hip; PNR_EXTRA_HIPCC_FLAGS="-DPNR_MARCH_WAVES=$w -DPNR_COOP_RAYS=$r" python -m palettenerf_amd.build >/dev/null 2>&1
  echo "waves $w rays $r: lego $(run) | $(run) ; garden $(run --workload garden --steps 20) ; palette $(run --workload lego_palette --steps 20)"
done; done
PNR_NO_COOP_MARCH=1; export PNR_NO_COOP_MARCH
echo "waves 5 nocoop: lego $(run) ; garden $(run --workload garden --steps 20) ; palette $(run --workload lego_palette --steps 20)"
