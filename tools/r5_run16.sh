#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "trains_like" -s 2>&1 | grep -E "fp32|passed|failed|Error" | cut -c1-700
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], json.dumps(d['roofline']['blocks_mfma_util']))"
VITRES_NTK=0 tools/prof_step.sh r5s_old --steps 20 --warmup 5 > /dev/null
for t in 1 2 3; do for b in 1 2 3; do
  VITRES_NTK_TILE=$t VITRES_NTK_BUF=$b tools/prof_step.sh r5s_t${t}b${b} --steps 20 --warmup 5 > /dev/null
done; done
tools/prof_step.sh r5s_auto --steps 20 --warmup 5 > /dev/null
