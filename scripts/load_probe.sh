#!/bin/bash
# the filter kernel with its probes taken out (measurement build, NEEDLE_NG_DBG=4: text loaded and waited for only), 4 and 8 units in flight
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
for rep in 1 2; do
for w in "$@"; do
for lib in needle_amd/libneedle_hip_tuning.so needle_amd/libneedle_hip_tuning_pf8.so; do
  for dbg in 4 1; do
    NEEDLE_PREFILTER_STRIDE=2 NEEDLE_LIB=$PWD/$lib NEEDLE_NG_DBG=$dbg timeout 300 python bench.py --workload $w --steps 20 --also none --no-cpu-baseline --no-extras --full-line 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $lib dbg $dbg', round(d['roofline']['kernel_ms'],4))" | tee -a gpurun_out/r6/load_probe.log
  done
done
done
done
