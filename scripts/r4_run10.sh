#!/bin/bash
# round 4: find-all behind the n-gram candidate filter: parity (three filter levels + the find-all suites), then C3-sparse timing
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_prefilter.py -x -q -m gpu > gpurun_out/r4/tests10.log 2>&1; grep -E "passed|failed" gpurun_out/r4/tests10.log | tail -2; grep -E "^E  |Error|assert" gpurun_out/r4/tests10.log | head -20
for e in "NEEDLE_FIND_ALL_FILTER=1" "NEEDLE_FIND_ALL_FILTER=0"; do
  echo "== c3s find-all $e"
  env $e python scripts/find_all_probe.py c3s 10000000 32 check 2>&1 | grep -v amdgpu | tail -1
  env $e FIND_ALL_PROBE_PACKED=1 python scripts/find_all_probe.py c3s 10000000 32 check 2>&1 | grep -v amdgpu | tail -1
done | tee gpurun_out/r4/find_all_filter_ab.log
