#!/bin/bash
# round 5, GPU run 1: wide kernel re-measured inside the step with the round-4 defaults; fixed-cost fit inputs (B = 32 / 64 / 128)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
EXP=vit-search_amd/lib/libvitres_hip_exp.so
export AB_STEPS=60 AB_WARMUP=20
tools/ab.sh "X=0" "VITRES_LIB=$EXP" "VITRES_LIB=$EXP VITRES_NT_WIDE=1" "VITRES_LIB=$EXP VITRES_NT_WIDE=1 VITRES_NTW_SK=0" "VITRES_LIB=$EXP VITRES_NT_WIDE=2" > gpurun_out/r5_wide_ab.txt 2>&1
for b in 32 64 128; do
  PROF_KEY=sr_tiny_supernet:$b:bf16 tools/prof_step.sh r5b$b --steps 20 --warmup 5 --batch $b
done
