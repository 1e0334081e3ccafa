#!/bin/bash
# round 4: where the tree stands -- GPU tests, driver-protocol + steady-state bench, kernel trace of the step
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_state.txt; : > $O
(timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r4_state_tests.txt
tail -3 gpurun_out/r4_state_tests.txt >> $O
echo "--- driver protocol (20 after 5)" >> $O
AB_STEPS=20 AB_WARMUP=5 tools/ab.sh "VITRES_X=0" >> $O 2>&1
echo "--- steady (100 after 30)" >> $O
AB_STEPS=100 AB_WARMUP=30 tools/ab.sh "VITRES_X=0" "VITRES_LN_BWD_LEAN=0" "VITRES_TAIL_AUX=0 VITRES_EMBED_WGRAD_SLICES=0" >> $O 2>&1
tools/prof_step.sh r4s --steps 20 --warmup 5 >> $O 2>&1
cat $O
