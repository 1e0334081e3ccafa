#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_e.txt; : > $O
export AB_STEPS=100 AB_WARMUP=30
tools/ab.sh "VITRES_TN_GROUP_CAP=20" "VITRES_TN_GROUP_CAP=25" "VITRES_TN_GROUP_CAP=30" "VITRES_TN_GROUP_CAP=18" "VITRES_TN_GROUP_CAP=20 VITRES_TN_S=48" "VITRES_TN_GROUP_CAP=20 VITRES_TN_S=24" "VITRES_TN_GROUP_CAP=20 VITRES_TN_S=16" >> $O 2>&1
echo "--- driver protocol" >> $O
AB_STEPS=20 AB_WARMUP=5 tools/ab.sh "VITRES_TN_GROUP_CAP=0" "VITRES_TN_GROUP_CAP=20" >> $O 2>&1
echo "--- per-step probe" >> $O
VITRES_DBG_STEPS=1 VITRES_TN_GROUP_CAP=20 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 5 2>&1 | grep "step probe" >> $O
cat $O
