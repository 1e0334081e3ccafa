#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
( timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r15_tests_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r15_smoke.txt 2>&1
( AB_STEPS=60 AB_WARMUP=20 tools/ab.sh "X=0" "VITRES_DBG_WGRAD8_MAXT=3000" "VITRES_DBG_WGRAD8_MAXT=9000" "VITRES_JOIN_LAG=1" "VITRES_JOIN_LAG=3" "VITRES_OPT_OVERLAP_BLOCKS=128" "VITRES_OPT_OVERLAP_BLOCKS=512" "X=0" 2>&1 ) > gpurun_out/r15_ab.txt
cat gpurun_out/r15_tests_all.txt gpurun_out/r15_smoke.txt gpurun_out/r15_ab.txt
