#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_find_all.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20
for rep in 1 2; do
for w in c3; do
  FIND_ALL_PROBE_DENSE_ONLY=1 python scripts/find_all_probe.py $w 10000000 32 check 2>/dev/null | tail -1
  FIND_ALL_PROBE_DENSE_ONLY=1 FIND_ALL_PROBE_PACKED=1 python scripts/find_all_probe.py $w 10000000 32 check 2>/dev/null | tail -1
  NEEDLE_LIB=$PWD/needle_amd/libneedle_hip_prev.so FIND_ALL_PROBE_DENSE_ONLY=1 python scripts/find_all_probe.py $w 10000000 32 check 2>/dev/null | tail -1
done; done
timeout 600 python scripts/fuzz_campaign.py 7000 60 2>&1 | tail -3
