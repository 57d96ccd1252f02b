#!/bin/bash
cd "$GRAFT_REPO_ROOT"
b() { timeout 300 python bench.py --workload $1 --steps 30 --also none --no-cpu-baseline --no-extras $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline'].get('frac'),4))"; }
for rep in 1 2; do
  NEEDLE_LIB=$PWD/needle_amd/libneedle_hip.so b c3s plain
  NEEDLE_LIB=$PWD/needle_amd/libneedle_hip_tuning_masked.so b c3s masked
  NEEDLE_LIB=$PWD/needle_amd/libneedle_hip.so b c3s plain_contained "--op contained_in"
  NEEDLE_LIB=$PWD/needle_amd/libneedle_hip_tuning_masked.so b c3s masked_contained "--op contained_in"
done
