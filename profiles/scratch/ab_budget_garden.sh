run() { python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['step_ms']['median'],3))"; }
for b in 1 2 3 4 8; do echo "budget $b: garden $(PNR_MARCH_BUDGET=$b run --workload garden) | $(PNR_MARCH_BUDGET=$b run --workload garden)   palette $(PNR_MARCH_BUDGET=$b run --workload lego_palette)"; done
for mb in 1280 2048; do echo "march_blocks $mb: garden $(PNR_MARCH_BLOCKS=$mb run --workload garden) | $(PNR_MARCH_BLOCKS=$mb run --workload garden)"; done
