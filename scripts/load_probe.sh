#!/bin/bash
# the filter kernel taken apart (measurement build, scripts/build_tuning.sh): NEEDLE_NG_DBG 0 whole, 1 candidates dropped, 2 candidates' text gathered but no
# walk, 4 no probes (text loaded and waited for only).  scripts/load_probe.sh "<levels>" workloads...
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
export NEEDLE_LIB=$PWD/needle_amd/libneedle_hip_tuning.so
L=$1; shift
for rep in 1 2; do
for w in "$@"; do
  for dbg in $L; do
    NEEDLE_NG_DBG=$dbg timeout 300 python bench.py --workload $w --steps 20 --also none --no-cpu-baseline --no-extras --full-line 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w dbg $dbg', round(d['roofline']['kernel_ms'],4))" | tee -a gpurun_out/r6/load_probe.log
  done
done
done
