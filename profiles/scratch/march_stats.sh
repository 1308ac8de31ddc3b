PNR_EXTRA_HIPCC_FLAGS="-DPNR_MARCH_STATS" python -m palettenerf_amd.build --force >/dev/null 2>&1
PNR_NO_COOP_MARCH=1 python profiles/march_stats.py 2>&1 | tail -4
PNR_NO_COOP_MARCH=1 python profiles/march_stats.py --workload garden 2>&1 | tail -3
