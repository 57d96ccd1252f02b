#!/bin/bash
# A/B of builds of the library inside ONE gpurun call (same box): scripts/ab.sh "<variant.so> ..." workloads...
cd "$GRAFT_REPO_ROOT"
VS=$1; shift
for w in "$@"; do
  for rep in 1 2; do
    for lib in needle_amd/libneedle_hip.so $VS; do
      NEEDLE_LIB=$PWD/$lib timeout 300 python bench.py --workload $w --steps 30 --also none --no-cpu-baseline --no-extras --full-line 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '$lib', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['achieved']))"
    done
  done
done
