#!/bin/bash
# round 4: filter kernel on strides that are not whole batches of KiB units (112, 208, 272 ...): parity, dictionary campaign (width 112 rows)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_prefilter.py -x -q -m gpu > gpurun_out/r4/tests12.log 2>&1; grep -E "passed|failed" gpurun_out/r4/tests12.log | tail -2; grep -E "^E  " gpurun_out/r4/tests12.log | head -8
FUZZ_MIN_LEN=5 python scripts/dictionary_fuzz.py 11000 24 > gpurun_out/r4/dictionary_fuzz_strides.log 2>&1; tail -1 gpurun_out/r4/dictionary_fuzz_strides.log; grep -c "filter stride" gpurun_out/r4/dictionary_fuzz_strides.log
NEEDLE_PREFILTER=1 python scripts/r4_ngram.py 2>&1 | grep -v amdgpu | tail -1
