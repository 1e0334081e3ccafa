#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "k_shares or deep_rings or lean_loop" 2>&1 | tail -15 ) > gpurun_out/r1_tests_kernels.txt
( timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "benched_configuration" 2>&1 | tail -15 ) > gpurun_out/r1_tests_oracle.txt
( timeout 600 python tools/ksplit_sweep.py s2 s3 2>&1 ) > gpurun_out/r1_sweep_dense.txt
( timeout 600 python tools/ksplit_sweep.py s2 s3 --masked 2>&1 ) > gpurun_out/r1_sweep_masked.txt
( AB_STEPS=60 AB_WARMUP=20 tools/ab.sh "VITRES_DBG_K_SHARES=1" "VITRES_DBG_K_SHARES=0" "VITRES_DBG_K_SHARES=1" "VITRES_DBG_K_SHARES=0" 2>&1 ) > gpurun_out/r1_ab.txt
python bench.py > gpurun_out/r1_bench.json 2> gpurun_out/r1_bench.err
tail -3 gpurun_out/r1_tests_kernels.txt gpurun_out/r1_tests_oracle.txt gpurun_out/r1_ab.txt
