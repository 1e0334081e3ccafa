#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for r in 0 2; do
  VITRES_NT_RING=$r tools/pmc_run.sh r5ring${r}_a "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" $PWD/tools/ring_bench.py skp
  VITRES_NT_RING=$r tools/pmc_run.sh r5ring${r}_b "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" $PWD/tools/ring_bench.py skp
  VITRES_NT_RING=$r tools/pmc_run.sh r5ring${r}_c "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum" $PWD/tools/ring_bench.py skp
done
grep -h "nt_kernel" gpurun_out/r5ring*_pmc.txt | cut -c1-60,81-160
tail -3 gpurun_out/r5ring0_c_pmc.log
