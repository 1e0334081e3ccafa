#!/bin/bash
# scratch: the command list of the last gpurun call of the round
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
P=$GRAFT_REPO_ROOT/gpurun_prev
( AB_STEPS=60 AB_WARMUP=20 tools/ab.sh "X=0" "VITRES_LIB=$P/lds128.so" "VITRES_LIB=$P/lds160.so" "X=0" "VITRES_LIB=$P/lds128.so" "VITRES_LIB=$P/lds160.so" 2>&1 ) > gpurun_out/r23_ab.txt
for wl in sr_small_supernet sr_tiny_mh_supernet ref_tiny; do for e in "X=0" "VITRES_LIB=$P/lds128.so"; do BENCH_FLAGS="--workload $wl" AB_STEPS=40 AB_WARMUP=10 tools/ab.sh "$e" | sed "s/^/$wl /"; done; done >> gpurun_out/r23_ab.txt 2>&1
cat gpurun_out/r23_ab.txt
