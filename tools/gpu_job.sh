#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
( timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "benched_configuration or ln_fold" 2>&1 | grep -v Warning | tail -12 ) > gpurun_out/r16_tests.txt
( timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_group" 2>&1 | tail -3 ) >> gpurun_out/r16_tests.txt
cat gpurun_out/r16_tests.txt
