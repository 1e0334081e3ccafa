#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_h.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "layernorm or ln_ or attention or attn" 2>&1 | tail -2 >> $O
export AB_STEPS=100 AB_WARMUP=30 VITRES_TN_GROUP_CAP=20
tools/ab.sh "VITRES_LN_XCD=0 VITRES_ATTN_XCD=0" "VITRES_LN_XCD=1 VITRES_ATTN_XCD=1" "VITRES_LN_XCD=0 VITRES_ATTN_XCD=0 VITRES_GROUP_INTERLEAVE=0" "VITRES_LN_XCD=1 VITRES_ATTN_XCD=1 VITRES_GROUP_INTERLEAVE=0" "VITRES_LN_XCD=1 VITRES_ATTN_XCD=0 VITRES_GROUP_INTERLEAVE=0" "VITRES_LN_XCD=0 VITRES_ATTN_XCD=1 VITRES_GROUP_INTERLEAVE=0" >> $O 2>&1
cat $O
