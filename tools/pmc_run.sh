#!/bin/bash
# usage (GPU box, repo root): tools/pmc_run.sh <tag> "<counters>" <python script + args>     -> gpurun_out/<tag>_pmc.txt
tag=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs -d $out -o t -- python "$@" > $root/gpurun_out/${tag}_pmc.log 2>&1
db=$(find $out -name '*.db' | head -1)
python $root/tools/rocpd_pmc.py $db > $root/gpurun_out/${tag}_pmc.txt 2>&1
rm -rf $out
