#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_step.sh <tag> [bench flags...]
# rocprofv3 kernel trace of a bench run; writes gpurun_out/<tag>_stats.txt (per-kernel table), <tag>_step.txt (one step in order)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out -o t -- python $root/bench.py --no-cpu-baseline --profile-steps 0 "$@" > $root/gpurun_out/${tag}_bench.log 2>&1
db=$(find $out -name '*.db' | head -1)
python $root/tools/rocpd_stats.py $db 60 > $root/gpurun_out/${tag}_stats.txt 2>&1
python $root/tools/rocpd_step.py $db 6 list > $root/gpurun_out/${tag}_step.txt 2>&1
python $root/tools/rocpd_queues.py $db > $root/gpurun_out/${tag}_queues.txt 2>&1
python $root/tools/rocpd_family_json.py $db $root/gpurun_out/${tag}_graph_kernels.json ${PROF_KEY:-sr_tiny_supernet:128:bf16} > /dev/null 2>&1
rm -rf $out
grep '^{' $root/gpurun_out/${tag}_bench.log | tail -1 | cut -c1-400
