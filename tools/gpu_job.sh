#!/bin/bash
# scratch: the command list of the last gpurun call of the round
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
for i in 1 2 3 4 5 6; do
  timeout 3000 python -X faulthandler -m pytest tests/ -x -v -m gpu > gpurun_out/r20_tests_$i.txt 2>&1
  echo "run $i rc=$?"; grep -n " passed\| failed" gpurun_out/r20_tests_$i.txt | tail -1 | cut -c1-200
  grep -n "Memory access\|fault\|Fatal\|Abort" gpurun_out/r20_tests_$i.txt | head -5
done
