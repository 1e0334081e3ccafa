#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_f.txt; : > $O
export VITRES_TN_GROUP_CAP=20
for ra in 0 2 3; do
echo "--- RUN_AHEAD=$ra per-step probe" >> $O
VITRES_RUN_AHEAD=$ra VITRES_DBG_STEPS=1 python bench.py --no-cpu-baseline --profile-steps 0 --steps 40 --warmup 5 2>&1 | grep "step probe" >> $O
done
echo "--- driver protocol" >> $O
AB_STEPS=20 AB_WARMUP=5 tools/ab.sh "VITRES_RUN_AHEAD=0" "VITRES_RUN_AHEAD=1" "VITRES_RUN_AHEAD=2" "VITRES_RUN_AHEAD=3" "VITRES_RUN_AHEAD=4" >> $O 2>&1
echo "--- steady" >> $O
AB_STEPS=100 AB_WARMUP=30 tools/ab.sh "VITRES_RUN_AHEAD=0" "VITRES_RUN_AHEAD=2" >> $O 2>&1
cat $O
