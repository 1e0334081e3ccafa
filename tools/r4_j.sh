#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_j.txt; : > $O
VITRES_NTLN_MI=2 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_ln" 2>&1 | tail -2 >> $O
export AB_STEPS=100 AB_WARMUP=30
tools/ab.sh "VITRES_NTLN_MI=4" "VITRES_NTLN_MI=2" >> $O 2>&1
E=$(pwd)/vit-search_amd/lib/libvitres_hip_exp.so
tools/ab.sh "VITRES_LIB=$E VITRES_LN_BWD_FORM=col" "VITRES_LIB=$E VITRES_LN_BWD_FORM=col VITRES_LN_BWD_R=4" >> $O 2>&1
echo "--- C5" >> $O
for e in "VITRES_NTLN_MI=4" "VITRES_NTLN_MI=2"; do for r in 1 2; do
 v=$(env $e python bench.py --workload evo_eval_sr_small --no-cpu-baseline --profile-steps 0 --steps 30 --warmup 10 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
 echo "$e: $v" >> $O; done; done
python tools/ntln_bench.py >> $O 2>&1
VITRES_NTLN_MI=2 python tools/ntln_bench.py >> $O 2>&1
cat $O
