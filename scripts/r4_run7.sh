#!/bin/bash
# round 4: the n-gram kernel's flood guard: parity (all three filter levels), the two hard texts and the bench text with the filter off / on
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_prefilter.py tests/test_gpu_find_packed16.py -x -q -m gpu > gpurun_out/r4/tests7.log 2>&1; grep -E "passed|failed|^E  " gpurun_out/r4/tests7.log | tail -6
for k in 0 1; do
  NEEDLE_PREFILTER=$k python scripts/r4_ngram.py 2>&1 | grep -v amdgpu | tail -1
  NEEDLE_PREFILTER=$k python scripts/r3_dense_dictionary.py 2>&1 | grep -v amdgpu | tail -2
  NEEDLE_PREFILTER=$k python scripts/r4_ngram_worstcase.py 2>&1 | grep -v amdgpu | tail -2
done | tee gpurun_out/r4/ngram_hard_texts2.log
