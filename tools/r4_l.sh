#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_l.txt; : > $O
export AB_STEPS=100 AB_WARMUP=30
tools/ab.sh "VITRES_LAST_UNCAP=0" "VITRES_LAST_UNCAP=1" "VITRES_LAST_UNCAP=1 VITRES_LAST_EARLY=1" "VITRES_LAST_UNCAP=0" "VITRES_LAST_UNCAP=1" >> $O 2>&1
cat $O | cut -c1-200
