#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_exp2.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "optimizer_inside or raises_mid_block" 2>&1 | grep -v "^  \|^$" | tail -40 >> $O
export AB_STEPS=100 AB_WARMUP=30
tools/ab.sh "VITRES_OPT_IN_GRAPH=0" "VITRES_OPT_IN_GRAPH=1 VITRES_OPT_RANGES=0" "VITRES_OPT_IN_GRAPH=1 VITRES_OPT_RANGE_BLOCKS=128" "VITRES_OPT_IN_GRAPH=1 VITRES_OPT_RANGE_BLOCKS=256" "VITRES_OPT_IN_GRAPH=1 VITRES_OPT_RANGE_BLOCKS=512" \
  "VITRES_OPT_IN_GRAPH=1 VITRES_OPT_RANGE_BLOCKS=1024" "VITRES_OPT_IN_GRAPH=1 VITRES_OPT_RANGE_BLOCKS=0" >> $O 2>&1
echo "--- driver protocol (20 after 5)" >> $O
export AB_STEPS=20 AB_WARMUP=5
tools/ab.sh "VITRES_OPT_IN_GRAPH=0" "VITRES_OPT_IN_GRAPH=1 VITRES_OPT_RANGE_BLOCKS=512" "VITRES_OPT_IN_GRAPH=1 VITRES_OPT_RANGE_BLOCKS=256" >> $O 2>&1
echo "--- per-step probe" >> $O
VITRES_DBG_STEPS=1 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 5 2>&1 | grep "step probe" >> $O
VITRES_DBG_STEPS=1 VITRES_OPT_IN_GRAPH=1 python bench.py --no-cpu-baseline --profile-steps 0 --steps 60 --warmup 5 2>&1 | grep "step probe" >> $O
cat $O
