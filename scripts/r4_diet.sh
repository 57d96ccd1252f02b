#!/bin/bash
# round 4, code diet of the big-table kernels: rolled piece loops (the build) against the unrolled form (libneedle_hip_tuning_unrolled.so,
# NEEDLE_EXTRA_DEFS=-DNEEDLE_BIG_ROLLED=0 NEEDLE_TUNING_SUFFIX=_unrolled scripts/build_tuning.sh) on the scan kernel's C3-sparse walk
# (NEEDLE_PREFILTER=0), full and ragged rows; instruction-fetch counters of both.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L 2>/dev/null | grep -i -E "IFETCH|ICACHE|INST_FETCH|SQC_" | head -40 > gpurun_out/r4/ifetch_counters.txt; head -30 gpurun_out/r4/ifetch_counters.txt
export NEEDLE_PREFILTER=0
for rep in 1 2; do
  for lib in needle_amd/libneedle_hip.so needle_amd/libneedle_hip_tuning_unrolled.so; do
    echo "== $lib"
    NEEDLE_LIB=$PWD/$lib timeout 300 python scripts/r4_ngram.py 2>&1 | grep -v amdgpu | tail -1
    NEEDLE_LIB=$PWD/$lib timeout 300 python scripts/quick_ragged_keywords.py 10000000 c3s 2>&1 | grep -v amdgpu
  done
done 2>&1 | tee gpurun_out/r4/diet_ab.log
for lib in needle_amd/libneedle_hip.so needle_amd/libneedle_hip_tuning_unrolled.so; do
  tag=$(basename $lib .so)
  NEEDLE_LIB=$PWD/$lib scripts/pmc.sh c3s diet_$tag "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_BUSY_CYCLES" > gpurun_out/r4/pmc_diet_$tag.log 2>&1
  tail -12 gpurun_out/r4/pmc_diet_$tag.log
done
