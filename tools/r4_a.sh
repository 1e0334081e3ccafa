#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_a.txt; : > $O
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "trains_like" 2>&1 | grep -v "^  \|^$" | tail -8 >> $O
echo "--- driver protocol (20 after 5)" >> $O
AB_STEPS=20 AB_WARMUP=5 tools/ab.sh "VITRES_X=0" "VITRES_TAIL_AUX=0 VITRES_EMBED_WGRAD_SLICES=0" >> $O 2>&1
echo "--- steady (100 after 30)" >> $O
AB_STEPS=100 AB_WARMUP=30 tools/ab.sh "VITRES_X=0" "VITRES_TAIL_AUX=0 VITRES_EMBED_WGRAD_SLICES=0" "VITRES_TAIL_AUX=0" "VITRES_DBG_SKIP_WGRAD=1" >> $O 2>&1
tools/prof_step.sh r4a --steps 20 --warmup 5 >> $O 2>&1
echo "--- C5" >> $O
for e in "VITRES_FUSED_MLP=0" "VITRES_FUSED_MLP=1" "VITRES_FUSED_MLP=2"; do for r in 1 2; do
 v=$(env $e python bench.py --workload evo_eval_sr_small --no-cpu-baseline --profile-steps 0 --steps 30 --warmup 10 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
 echo "$e: $v" >> $O; done; done
cat $O
