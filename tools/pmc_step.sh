#!/bin/bash
# usage (GPU box, repo root): tools/pmc_step.sh <tag>  -> gpurun_out/<tag>_{g,h}_pmc.txt : instruction mix / LDS conflicts and L2 hit
# rates per kernel of one single-stream eager step (separate PMC passes, kernel trace only)
tag=$1
B="bench.py --steps 1 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-graph"
VITRES_OVERLAP=0 tools/pmc_run.sh ${tag}_g "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES" $(pwd)/$B
VITRES_OVERLAP=0 tools/pmc_run.sh ${tag}_h "TCC_HIT_sum TCC_MISS_sum" $(pwd)/$B
