#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_exp5.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused_mlp or layernorm" 2>&1 | grep -v "^  \|^$" | tail -30 >> $O
VITRES_LN_BWD_LEAN=2 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "layernorm" 2>&1 | tail -3 >> $O
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "trains_like or micro_supernet or fullsize" 2>&1 | grep -v "^  \|^$" | tail -12 >> $O
echo "--- C5" >> $O
for e in "VITRES_FUSED_MLP=0" "VITRES_FUSED_MLP=1"; do for r in 1 2; do
 v=$(env $e python bench.py --workload evo_eval_sr_small --no-cpu-baseline --profile-steps 0 --steps 30 --warmup 10 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
 echo "$e: $v" >> $O; done; done
bash tools/prof_evo.sh r4_evo_fused > /dev/null 2>&1
head -14 gpurun_out/r4_evo_fused_stats.txt | cut -c1-150 >> $O
export AB_STEPS=100 AB_WARMUP=30
tools/ab.sh "VITRES_LN_BWD_LEAN=0" "VITRES_LN_BWD_LEAN=1" "VITRES_LN_BWD_LEAN=2" "VITRES_LN_BWD_LEAN=1 VITRES_TN_GROUP_FILL=1 VITRES_TN_S=64" "VITRES_LN_BWD_LEAN=0 VITRES_TN_GROUP_FILL=1 VITRES_TN_S=64" >> $O 2>&1
cat $O
