#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for r in 0 1 2 3 4; do VITRES_NT_RING=$r python tools/ring_bench.py > gpurun_out/r5_ring_$r.txt 2>&1; done
paste gpurun_out/r5_ring_0.txt <(cut -c29-48 gpurun_out/r5_ring_1.txt) <(cut -c29-48 gpurun_out/r5_ring_2.txt) <(cut -c29-48 gpurun_out/r5_ring_3.txt) <(cut -c29-48 gpurun_out/r5_ring_4.txt) > gpurun_out/r5_ring_table.txt
export AB_STEPS=60 AB_WARMUP=20
tools/ab.sh "VITRES_NT_RING=0" "VITRES_NT_RING=1" "VITRES_NT_RING=2" "VITRES_NT_RING=3" "VITRES_NT_RING=4" > gpurun_out/r5_ring_ab.txt 2>&1
cat gpurun_out/r5_ring_table.txt gpurun_out/r5_ring_ab.txt
