#!/bin/bash
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out -o t -- python $root/bench.py --workload evo_eval_sr_small --steps 24 --warmup 4 --profile-steps 0 "$@" > $root/gpurun_out/${tag}_bench.log 2>&1
db=$(find $out -name '*.db' | head -1)
python $root/tools/rocpd_stats.py $db 40 > $root/gpurun_out/${tag}_stats.txt 2>&1
rm -rf $out
grep '^{' $root/gpurun_out/${tag}_bench.log | tail -1 | cut -c1-300
