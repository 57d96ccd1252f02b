#!/bin/bash
cd "$GRAFT_REPO_ROOT"
N=${1:-10000000}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_find_all.py -m gpu -x -q > gpurun_out/fa_tests.log 2>&1
tail -3 gpurun_out/fa_tests.log
for w in c3 c2 c5 c3s; do
  timeout 300 python scripts/find_all_probe.py $w $N 32 2>&1 | tail -1
  NEEDLE_DEBUG_NO_BACKWARD=1 timeout 300 python scripts/find_all_probe.py $w $N 32 2>&1 | tail -1
done
