#!/bin/bash
# find-all: parity tests, then the one-pass kernel's time on the bench batches (scripts/find_all_probe.py)
cd "$GRAFT_REPO_ROOT"
N=${1:-10000000}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_find_all.py tests/test_gpu_configs.py tests/test_gpu_packed.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -m gpu -x -q -k "find_all or every_match or nullable or packed or dictionary or random_regexes or no_slots" > gpurun_out/fa_tests.log 2>&1
tail -3 gpurun_out/fa_tests.log
for w in c3 c2 c5 c3s; do
  timeout 300 python scripts/find_all_probe.py $w $N 32 2>&1 | tail -1
  NEEDLE_FIND_ALL_DEFER=0 FIND_ALL_PROBE_DENSE_ONLY=1 timeout 300 python scripts/find_all_probe.py $w $N 32 2>&1 | tail -1
done
