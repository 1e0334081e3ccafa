#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "lean_loop or (k_shares and 2176-3072)" 2>&1 | tail -4 ) > gpurun_out/r7_tests_kernels.txt
( AB_STEPS=60 AB_WARMUP=20 tools/ab.sh "X=0" "VITRES_LIB=$GRAFT_REPO_ROOT/gpurun_prev/libvitres_hip.so" "X=0" "VITRES_LIB=$GRAFT_REPO_ROOT/gpurun_prev/libvitres_hip.so" 2>&1 ) > gpurun_out/r7_ab.txt
( timeout 600 python tools/ksplit_sweep.py s2 s3 2>&1 | cut -c1-50 ) > gpurun_out/r7_sweep_auto.txt
cat gpurun_out/r7_tests_kernels.txt gpurun_out/r7_ab.txt gpurun_out/r7_sweep_auto.txt
