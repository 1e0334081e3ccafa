#!/bin/bash
# round 4, experiment 1: step tail on auxiliary streams + AdamW range updates beside the backward
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_exp1.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "optimizer_inside or hipgraph or flat_adamw" 2>&1 | tail -5 >> $O
export AB_STEPS=100 AB_WARMUP=30
tools/ab.sh "VITRES_TAIL_AUX=0 VITRES_EMBED_WGRAD_SLICES=0" "VITRES_TAIL_AUX=0 VITRES_EMBED_WGRAD_SLICES=8" "VITRES_TAIL_AUX=1 VITRES_EMBED_WGRAD_SLICES=0" \
  "VITRES_TAIL_AUX=1" "VITRES_TAIL_AUX=1 VITRES_EMBED_WGRAD_SLICES=4" \
  "VITRES_OPT_IN_GRAPH=1 VITRES_OPT_RANGES=0" "VITRES_OPT_IN_GRAPH=1 VITRES_OPT_RANGE_BLOCKS=256" "VITRES_OPT_IN_GRAPH=1 VITRES_OPT_RANGE_BLOCKS=512" \
  "VITRES_OPT_IN_GRAPH=1 VITRES_OPT_RANGE_BLOCKS=1024" "VITRES_OPT_IN_GRAPH=1 VITRES_OPT_RANGE_BLOCKS=2048" "VITRES_OPT_IN_GRAPH=1 VITRES_OPT_RANGE_BLOCKS=0" >> $O 2>&1
echo "--- driver protocol (20 after 5)" >> $O
export AB_STEPS=20 AB_WARMUP=5
tools/ab.sh "VITRES_TAIL_AUX=0 VITRES_EMBED_WGRAD_SLICES=0" "VITRES_TAIL_AUX=1" "VITRES_OPT_IN_GRAPH=1 VITRES_OPT_RANGE_BLOCKS=512" >> $O 2>&1
cat $O
