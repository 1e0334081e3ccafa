#!/bin/bash
# usage (GPU box, repo root): tools/traffic_run.sh <tag> <workload> <batch> [bench flags]  -> gpurun_out/<tag>_traffic.txt / .json
# (copy the .json to profiles/r04_traffic_<workload>.json: bench.py reads it when workload, per-GPU batch and dtype match)
# two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE; separate, kernel-trace only) over a single-stream eager bench run
tag=$1; wl=$2; batch=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export VITRES_OVERLAP=0
for c in FETCH_SIZE WRITE_SIZE; do
  out=$root/gpurun_out/pmc_${tag}_$c; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --pmc $c -d $out -o t -- python $root/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --profile-steps 0 --workload $wl --batch $batch "$@" > $root/gpurun_out/${tag}_$c.log 2>&1
done
f=$(find $root/gpurun_out/pmc_${tag}_FETCH_SIZE -name '*.db' | head -1)
w=$(find $root/gpurun_out/pmc_${tag}_WRITE_SIZE -name '*.db' | head -1)
python $root/tools/rocpd_traffic.py $f $w $root/gpurun_out/${tag}_traffic.json $wl:$batch:bf16 > $root/gpurun_out/${tag}_traffic.txt 2>&1
rm -rf $root/gpurun_out/pmc_${tag}_FETCH_SIZE $root/gpurun_out/pmc_${tag}_WRITE_SIZE
