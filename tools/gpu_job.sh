#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
( AB_STEPS=60 AB_WARMUP=20 tools/ab.sh "X=0" "VITRES_LIB=$GRAFT_REPO_ROOT/gpurun_prev/libvitres_hip.so" "X=0" "VITRES_LIB=$GRAFT_REPO_ROOT/gpurun_prev/libvitres_hip.so" "X=0" "VITRES_LIB=$GRAFT_REPO_ROOT/gpurun_prev/libvitres_hip.so" 2>&1 ) > gpurun_out/r11_ab.txt
cat gpurun_out/r11_ab.txt
