#!/bin/bash
# scratch: the command list of the last gpurun call of the round
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
P=$GRAFT_REPO_ROOT/gpurun_prev
( AB_STEPS=60 AB_WARMUP=20 tools/ab.sh "X=0" "VITRES_LIB=$P/cap15.so" "VITRES_LIB=$P/cap25.so" "VITRES_LIB=$P/cap30.so" "X=0" "VITRES_LIB=$P/cap15.so" "VITRES_LIB=$P/cap25.so" "VITRES_LIB=$P/cap30.so" 2>&1 ) > gpurun_out/r21_ab.txt
cat gpurun_out/r21_ab.txt
