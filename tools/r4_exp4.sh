#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_exp4.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused_mlp" 2>&1 | grep -v "^  \|^$" | tail -30 >> $O
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "optimizer_inside or raises_mid_block or c5 or evo" 2>&1 | tail -5 >> $O
export AB_STEPS=100 AB_WARMUP=30
tools/ab.sh "VITRES_TN_S=32" "VITRES_TN_GROUP_FILL=1 VITRES_TN_S=64" "VITRES_TN_GROUP_FILL=1 VITRES_TN_S=80" "VITRES_TN_GROUP_FILL=1 VITRES_TN_S=96" "VITRES_TN_GROUP_FILL=1 VITRES_TN_S=128" "VITRES_TN_GROUP_FILL=1 VITRES_TN_S=64 VITRES_JOIN_LAG=3" >> $O 2>&1
echo "--- C5" >> $O
for e in "VITRES_FUSED_MLP=0" "VITRES_FUSED_MLP=1"; do for r in 1 2; do
 v=$(env $e python bench.py --workload evo_eval_sr_small --no-cpu-baseline --profile-steps 0 --steps 30 --warmup 10 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
 echo "$e: $v" >> $O; done; done
cat $O
