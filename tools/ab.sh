#!/bin/bash
# usage: tools/ab.sh "ENV1=.. ENV2=.." "ENV=.." ...   -> ms_per_step of bench.py for each environment (two runs each)
for e in "$@"; do
  for r in 1 2; do
    v=$(env $e python bench.py --no-cpu-baseline --profile-steps 0 --steps ${AB_STEPS:-100} --warmup ${AB_WARMUP:-30} $BENCH_FLAGS 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['host_ms_per_step'])")
    echo "$e: $v"
  done
done
