#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root && mkdir -p gpurun_out
O=$root/gpurun_out/r4_c.txt; : > $O
out=$root/gpurun_out/prof_2c; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out -o t -- python $root/tools/two_chain_probe.py > $root/gpurun_out/r4_c_probe.log 2>&1
db=$(find $out -name '*.db' | head -1)
grep "graph" $root/gpurun_out/r4_c_probe.log >> $O
python $root/tools/rocpd_conc.py $db 40 >> $O 2>&1
rm -rf $out
cd $root
VITRES_LN_BWD_LEAN=3 tools/prof_step.sh r4c_col --steps 20 --warmup 5 >> $O 2>&1
VITRES_DBG_SKIP_WGRAD=1 tools/prof_step.sh r4c_nowg --steps 20 --warmup 5 >> $O 2>&1
cat $O | cut -c1-300
