#!/bin/bash
# round 4: find-all's lengths program with window addressing (NEEDLE_FIND_ALL_WINDOW=1, the build's default) against column maps
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_find_all.py -x -q -m gpu > gpurun_out/r4/tests9.log 2>&1; grep -E "passed|failed|^E  " gpurun_out/r4/tests9.log | tail -4
for rep in 1 2; do for wnd in 1 0; do
  echo "== NEEDLE_FIND_ALL_WINDOW=$wnd"
  NEEDLE_FIND_ALL_WINDOW=$wnd python scripts/find_all_probe.py c3 10000000 32 check 2>&1 | grep -v amdgpu | tail -1
  NEEDLE_FIND_ALL_WINDOW=$wnd FIND_ALL_PROBE_PACKED=1 python scripts/find_all_probe.py c3 10000000 32 check 2>&1 | grep -v amdgpu | tail -1
done; done | tee gpurun_out/r4/find_all_window_ab.log
