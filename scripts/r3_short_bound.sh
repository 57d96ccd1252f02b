#!/bin/bash
# short rows: what find() costs without its backward walks (tuning build: start := end) -- the bound for a better backward walk
cd "$GRAFT_REPO_ROOT"
export NEEDLE_LIB=$PWD/needle_amd/libneedle_hip_tuning.so
for s in 16 32 64; do
  echo "== stride $s"; python scripts/short_rows_rate.py $s 2>&1 | grep -v amdgpu
  echo "-- no backward"; NEEDLE_DEBUG_NO_BACKWARD=1 python scripts/short_rows_rate.py $s 2>&1 | grep find
done
