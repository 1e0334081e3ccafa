#!/bin/bash
# usage (GPU box, repo root): tools/final_profiles.sh <round tag, e.g. r02>   -> gpurun_out/<tag>_*  (copy the summaries to profiles/)
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
# (a) kernel-trace summary of the default bench command (hipGraph, two streams)
out=$root/gpurun_out/prof_${tag}_a; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o t -- python $root/bench.py --no-cpu-baseline > $root/gpurun_out/${tag}_a_bench.log 2>&1
db=$(find $out -name '*.db' | head -1)
python $root/tools/rocpd_stats.py $db 60 > $root/gpurun_out/${tag}_a_kernel_stats.txt 2>&1
python $root/tools/rocpd_step.py $db 6 > $root/gpurun_out/${tag}_a_step.txt 2>&1
rm -rf $out
# (d) HBM traffic per kernel family (FETCH_SIZE / WRITE_SIZE, separate passes, single stream, eager)
$root/tools/traffic_run.sh ${tag}_d
# (f) MFMA-busy / wave states (single stream, eager)
out=$root/gpurun_out/prof_${tag}_f; rm -rf $out; mkdir -p $out
cd /tmp
VITRES_OVERLAP=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out -o t -- python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-graph > $root/gpurun_out/${tag}_f.log 2>&1
db=$(find $out -name '*.db' | head -1)
python $root/tools/rocpd_mfma.py $db > $root/gpurun_out/${tag}_f_mfma_busy_pmc.txt 2>&1
rm -rf $out
# (e) C5: HBM traffic of the candidate-scoring forward
$root/tools/traffic_run.sh ${tag}_evo --workload evo_eval_sr_small
# (g) per-shape table of every GEMM launch, kernels alone (single stream)
cd $root
VITRES_OVERLAP=0 python bench.py --no-cpu-baseline --launch-table gpurun_out/${tag}_launch_table.txt > /dev/null 2>&1
# (h) the gate-meeting path: the same step on the exact-fp32 kernels
python bench.py --dtype f32 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${tag}_bench_c3_sr_tiny_f32.json
# bench lines of the other configurations
cd $root
python bench.py > gpurun_out/${tag}_bench_c3_sr_tiny.json 2> gpurun_out/${tag}_bench_c3.err
python bench.py --workload ref_tiny --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${tag}_bench_c2_ref_tiny.json
python bench.py --workload sr_tiny_mh_supernet --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${tag}_bench_c3p_sr_tiny_mh.json
python bench.py --workload sr_small_supernet --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${tag}_bench_c4_sr_small.json
python bench.py --workload evo_eval_sr_small --steps 64 --warmup 8 2>/dev/null | grep '^{' > gpurun_out/${tag}_bench_c5_evo_eval.json
for f in c3_sr_tiny c3_sr_tiny_f32 c2_ref_tiny c3p_sr_tiny_mh c4_sr_small c5_evo_eval; do python -c "
import json,sys
l=[x for x in open('gpurun_out/${tag}_bench_$f.json') if x.startswith('{')]
d=json.loads(l[-1]); print('$f', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))"; done
