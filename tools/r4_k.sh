#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_k.txt; : > $O
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3 >> $O
export AB_STEPS=100 AB_WARMUP=30
tools/ab.sh "VITRES_LAST_EARLY=0" "VITRES_LAST_EARLY=1" "VITRES_LAST_EARLY=0" "VITRES_LAST_EARLY=1" >> $O 2>&1
echo "--- driver" >> $O
AB_STEPS=20 AB_WARMUP=5 tools/ab.sh "VITRES_LAST_EARLY=0" "VITRES_LAST_EARLY=1" >> $O 2>&1
tools/prof_step.sh r4k --steps 20 --warmup 5 >> $O 2>&1
tail -14 gpurun_out/r4k_step.txt | cut -c1-120 >> $O
cat $O | cut -c1-200
