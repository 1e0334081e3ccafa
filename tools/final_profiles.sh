#!/bin/bash
# usage (GPU box, repo root): tools/final_profiles.sh <round tag, e.g. r04>   -> gpurun_out/<tag>_*  (copy the summaries to profiles/)
# Everything profiles/<tag>_* is made of.  Protocol of every bench line: the driver's (--steps 20 --warmup 5); one steady-state
# line (--steps 100 --warmup 30) of the headline workload beside it.
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root && mkdir -p gpurun_out profiles
batch_of() { python - "$1" <<'PY'
import sys, re
src = open("bench.py").read()
print(re.search(r'"%s": dict\(.*?batch=(\d+)' % sys.argv[1], src).group(1))
PY
}
# (a) kernel trace of the default bench command (hipGraph, two streams): per-kernel table, one step in order, queues, in-graph averages
PROF_KEY=sr_tiny_supernet:128:bf16 tools/prof_step.sh ${tag}_a --steps 20 --warmup 5 > /dev/null 2>&1
cp gpurun_out/${tag}_a_graph_kernels.json profiles/${tag}_graph_kernels_sr_tiny_supernet.json
# (d) HBM traffic per kernel family (FETCH_SIZE / WRITE_SIZE, separate passes, single stream, eager) of every workload
for wl in sr_tiny_supernet ref_tiny sr_tiny_mh_supernet sr_small_supernet evo_eval_sr_small; do
  b=$(batch_of $wl)
  tools/traffic_run.sh ${tag}_$wl $wl $b
  cp gpurun_out/${tag}_${wl}_traffic.json profiles/${tag}_traffic_$wl.json
done
# (f) MFMA-busy / wave states (single stream, eager)
out=$root/gpurun_out/prof_${tag}_f; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
VITRES_OVERLAP=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out -o t -- python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-graph > $root/gpurun_out/${tag}_f.log 2>&1
db=$(find $out -name '*.db' | head -1)
python $root/tools/rocpd_mfma.py $db > $root/gpurun_out/${tag}_f_mfma_busy_pmc.txt 2>&1
rm -rf $out
cd $root
# (g) per-shape table of every GEMM launch, kernels alone (single stream)
VITRES_OVERLAP=0 python bench.py --no-cpu-baseline --launch-table gpurun_out/${tag}_launch_table.txt > /dev/null 2>&1
# (h) bench lines, the driver's protocol; cpu_baseline and traffic on every one
python bench.py > gpurun_out/${tag}_bench_c3_sr_tiny.json 2> gpurun_out/${tag}_bench_c3.err
python bench.py --steps 100 --warmup 30 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${tag}_bench_c3_sr_tiny_steady.json
python bench.py --dtype f32 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${tag}_bench_c3_sr_tiny_f32.json
python bench.py --workload ref_tiny 2>/dev/null | grep '^{' > gpurun_out/${tag}_bench_c2_ref_tiny.json
python bench.py --workload sr_tiny_mh_supernet 2>/dev/null | grep '^{' > gpurun_out/${tag}_bench_c3p_sr_tiny_mh.json
python bench.py --workload sr_small_supernet 2>/dev/null | grep '^{' > gpurun_out/${tag}_bench_c4_sr_small.json
python bench.py --workload evo_eval_sr_small --steps 64 --warmup 8 2>/dev/null | grep '^{' > gpurun_out/${tag}_bench_c5_evo_eval.json
for f in c3_sr_tiny c3_sr_tiny_steady c3_sr_tiny_f32 c2_ref_tiny c3p_sr_tiny_mh c4_sr_small c5_evo_eval; do python -c "
import json,sys
l=[x for x in open('gpurun_out/${tag}_bench_$f.json') if x.startswith('{')]
d=json.loads(l[-1]); r=d.get('roofline') or {}; print('$f', d['value'], d['ms_per_step'], r.get('frac'), r.get('traffic'), (d.get('cpu_baseline') or {}).get('value'))"; done
