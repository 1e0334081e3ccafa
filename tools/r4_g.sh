#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r4_g.txt; : > $O
hipcc --offload-arch=gfx950 -O3 tools/probes/xcd_reuse_probe.hip -o /tmp/xcd_probe 2>/dev/null && /tmp/xcd_probe >> $O 2>&1
echo "--- gap probe" >> $O
VITRES_DBG_GAP=1 python bench.py --no-cpu-baseline --profile-steps 0 --steps 40 --warmup 10 2>&1 | grep "gap probe" >> $O
VITRES_DBG_GAP=1 VITRES_OPT_IN_GRAPH=1 python bench.py --no-cpu-baseline --profile-steps 0 --steps 40 --warmup 10 2>&1 | grep "gap probe" >> $O
cat $O
