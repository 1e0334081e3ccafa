#!/bin/bash
# re-tune of the step's knobs at the current kernel speeds (two runs each): tools/sweep_knobs.sh > gpurun_out/<tag>.txt
bash tools/ab.sh "X=0" "VITRES_TN_S=16" "VITRES_TN_S=24" "VITRES_TN_S=48" "VITRES_TN_GROUP_FILL=1" "VITRES_TN_GROUP_FILL=3" "VITRES_TN_GROUP_FILL=4" \
  "VITRES_JOIN_LAG=1" "VITRES_JOIN_LAG=3" "VITRES_FUSE_LN=1" "VITRES_FUSE_LN=2" "VITRES_NT_PAIR=0" "VITRES_SIDE_STREAMS=2" "VITRES_TN_STAGES=3" \
  "VITRES_LN_COPIES=16" "VITRES_ATTN_BWD_SHORT=2" "VITRES_TN_FILL=2" "VITRES_TN_FILL=6" "X=1"
