#!/bin/bash
# round 4: the flood watch of the n-gram filter kernel: parity, then the decapitated-keyword text with the watch on / off and the bench text
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_prefilter.py -x -q -m gpu > gpurun_out/r4/tests8.log 2>&1; grep -E "passed|failed|^E  " gpurun_out/r4/tests8.log | tail -4
NEEDLE_PREFILTER=0 python scripts/r4_ngram_worstcase.py 2>&1 | grep -v amdgpu | tail -3
NEEDLE_PREFILTER=1 python scripts/r4_ngram_worstcase.py 2>&1 | grep -v amdgpu | tail -3
NEEDLE_PREFILTER=1 NEEDLE_PREFILTER_WATCH=0 python scripts/r4_ngram_worstcase.py 2>&1 | grep -v amdgpu | tail -3
NEEDLE_PREFILTER=1 python scripts/r4_ngram.py 2>&1 | grep -v amdgpu | tail -1
NEEDLE_PREFILTER=1 python scripts/r3_dense_dictionary.py 2>&1 | grep -v amdgpu | tail -2
