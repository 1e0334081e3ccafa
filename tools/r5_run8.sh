#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export AB_STEPS=60 AB_WARMUP=20
tools/ab.sh "VITRES_NTK=0" "VITRES_NTK=1" "VITRES_NTK=1 VITRES_NTK_BUF=1" "VITRES_NTK=1 VITRES_NTK_BUF=2" "VITRES_NTK=1 VITRES_NTK_BUF=3" 2>&1 | grep -v amdgpu
for k in 0 1; do VITRES_NTK=$k python bench.py --no-cpu-baseline --steps 10 --warmup 5 --launch-table gpurun_out/r5_lt_ntk$k.txt > /dev/null 2>&1; done
