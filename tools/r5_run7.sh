#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
VITRES_NTK=0 python tools/ring_bench.py kprobe 2>&1 | grep -v amdgpu > gpurun_out/r5_kp_old.txt
for t in 1 2; do for b in 1 2 3; do
  VITRES_NTK_TILE=$t VITRES_NTK_BUF=$b python tools/ring_bench.py kprobe 2>&1 | grep -v amdgpu > gpurun_out/r5_kp_t${t}b${b}.txt
done; done
paste gpurun_out/r5_kp_old.txt <(cut -c29-40 gpurun_out/r5_kp_t1b1.txt) <(cut -c29-40 gpurun_out/r5_kp_t1b2.txt) <(cut -c29-40 gpurun_out/r5_kp_t1b3.txt) <(cut -c29-40 gpurun_out/r5_kp_t2b1.txt) <(cut -c29-40 gpurun_out/r5_kp_t2b2.txt) <(cut -c29-40 gpurun_out/r5_kp_t2b3.txt)
