#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_i.txt; : > $O
(timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -12) > gpurun_out/r4_i_tests.txt
tail -3 gpurun_out/r4_i_tests.txt >> $O
(VITRES_LIB=$(pwd)/vit-search_amd/lib/libvitres_hip_exp.so timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wide or gemm_ln or fused_mlp" 2>&1 | tail -3) >> $O
echo "--- driver protocol" >> $O
AB_STEPS=20 AB_WARMUP=5 tools/ab.sh "VITRES_X=0" >> $O 2>&1
echo "--- steady" >> $O
AB_STEPS=100 AB_WARMUP=30 tools/ab.sh "VITRES_X=0" >> $O 2>&1
cat $O
