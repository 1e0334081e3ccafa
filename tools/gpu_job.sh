#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "ln_fold" 2>&1 | tail -3 ) > gpurun_out/r10_tests_kernels.txt
( AB_STEPS=60 AB_WARMUP=20 tools/ab.sh "X=0" "VITRES_LN_FOLD=0" "X=0" "VITRES_LN_FOLD=0" 2>&1 | grep -v "^  File\|^Trace\|^    " ) > gpurun_out/r10_ab.txt
VITRES_OVERLAP=0 python bench.py --no-cpu-baseline --launch-table gpurun_out/r10_launch_table.txt > /dev/null 2>&1
tail -3 gpurun_out/r10_tests_kernels.txt; cat gpurun_out/r10_ab.txt; grep "nt+ln\|res scale f32out" gpurun_out/r10_launch_table.txt
