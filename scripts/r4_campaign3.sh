#!/bin/bash
# round 4: the filter kernels (find, containedIn, find-all) under every pattern the analysis accepts: NEEDLE_PREFILTER=2 with plain tables
# (no pair table) over the regex fuzz campaign, the fuzz / find-all / parity suites and the dictionary campaign
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
export NEEDLE_PREFILTER=2 NEEDLE_PAIR_MAX_BYTES=0
python scripts/fuzz_campaign.py 7000 ${1:-200} > gpurun_out/r4/fuzz_campaign_level2.log 2>&1; tail -2 gpurun_out/r4/fuzz_campaign_level2.log
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_find_all.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_real_text.py tests/test_gpu_matches_txt_batch.py tests/test_gpu_packed.py tests/test_gpu_find_packed16.py tests/test_gpu_compact.py -q -m gpu > gpurun_out/r4/tests_level2.log 2>&1; grep -E "passed|failed" gpurun_out/r4/tests_level2.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/r4/tests_level2.log | head -20
FUZZ_MIN_LEN=5 python scripts/dictionary_fuzz.py 8000 ${2:-32} > gpurun_out/r4/dictionary_fuzz_min5_level2b.log 2>&1; tail -1 gpurun_out/r4/dictionary_fuzz_min5_level2b.log
