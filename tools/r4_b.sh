#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_b.txt; : > $O
timeout 600 python tools/two_chain_probe.py 2>&1 | grep -v Warn | tail -6 >> $O
for L in 3; do for R in 2 4; do
VITRES_LN_BWD_LEAN=$L VITRES_LN_BWD_R=$R timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "layernorm or ln_" 2>&1 | tail -3 >> $O
done; done
AB_STEPS=100 AB_WARMUP=30 tools/ab.sh "VITRES_LN_BWD_LEAN=0" "VITRES_LN_BWD_LEAN=3" "VITRES_LN_BWD_LEAN=3 VITRES_LN_BWD_R=2" "VITRES_LN_BWD_LEAN=3 VITRES_LN_BWD_R=4" "VITRES_LN_BWD_LEAN=4 VITRES_LN_BWD_R=2" >> $O 2>&1
python tools/ln_bench.py >> $O 2>&1
VITRES_LN_BWD_LEAN=3 python tools/ln_bench.py >> $O 2>&1
cat $O
