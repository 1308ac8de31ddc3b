#!/bin/bash
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_final3.log 2>&1; echo "rc $?" >> $O/pytest_final3.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log
