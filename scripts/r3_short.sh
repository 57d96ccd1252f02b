#!/bin/bash
# round 3, item 7: find() on short rows -- the unguarded walk + LDS-window backward walk against the previous build, same box
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_packed.py tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_reference_asserts.py tests/test_gpu_matches_txt_batch.py -x -q 2>&1 | tail -3
for stride in 16 32 64; do
  for lib in needle_amd/libneedle_hip_prev.so needle_amd/libneedle_hip.so; do
    echo "== $lib"
    NEEDLE_LIB=$PWD/$lib python scripts/short_rows_rate.py $stride 2>/dev/null | tail -2
  done
done
for lib in needle_amd/libneedle_hip_prev.so needle_amd/libneedle_hip.so; do
  echo "== $lib keywords on 64-byte rows"
  NEEDLE_LIB=$PWD/$lib python scripts/short_rows_rate.py 64 "Sherlock|Holmes|Watson|[0-9]+x" 2>/dev/null | tail -2
done
