#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_d.txt; : > $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "group or wgrad" 2>&1 | tail -2 >> $O
VITRES_TN_GROUP_CAP=10 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "group or wgrad" 2>&1 | tail -2 >> $O
export AB_STEPS=100 AB_WARMUP=30
tools/ab.sh "VITRES_X=0" "VITRES_TN_GROUP_CAP=20" "VITRES_TN_GROUP_CAP=15" "VITRES_TN_GROUP_CAP=10" "VITRES_TN_GROUP_CAP=10 VITRES_JOIN_LAG=4" "VITRES_TN_GROUP_CAP=15 VITRES_JOIN_LAG=4" \
  "VITRES_TN_GROUP_FILL=2 VITRES_TN_S=64" "VITRES_TN_GROUP_FILL=2 VITRES_TN_S=128" "VITRES_TN_GROUP_FILL=1 VITRES_TN_S=128" "VITRES_TN_GROUP_FILL=1 VITRES_TN_S=128 VITRES_JOIN_LAG=4" >> $O 2>&1
cat $O
