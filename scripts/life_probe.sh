#!/bin/bash
# scripts/life_probe.sh "<dbg levels>" workloads...: section shares + wave lifetimes of the filter kernel (measurement build)
cd "$GRAFT_REPO_ROOT"
export NEEDLE_LIB=$PWD/needle_amd/libneedle_hip_tuning.so
mkdir -p gpurun_out/r6
DBGS=$1; shift
for w in "$@"; do
  for dbg in $DBGS; do
  echo "== $w dbg $dbg" | tee -a gpurun_out/r6/life_probe.log
  NEEDLE_PREFILTER_STRIDE=2 NEEDLE_NG_DBG=$dbg NEEDLE_NG_STAMPS=1 timeout 300 python bench.py --workload $w --steps 2 --warmup 1 --also none --no-cpu-baseline --no-extras --full-line 2>&1 | grep -E "NG-STAMPS|NG-LIFE" | tail -2 | tee -a gpurun_out/r6/life_probe.log
  done
done
