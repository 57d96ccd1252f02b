#!/bin/bash
# round 4: random dictionaries incl. find-all (dense slots, count pass) -- with filters (FUZZ_MIN_LEN=5 / 7, default level and level 2) and
# without (the one-pass kernel)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
FUZZ_MIN_LEN=5 python scripts/dictionary_fuzz.py 9000 ${1:-40} > gpurun_out/r4/dictionary_fuzz_fa_min5.log 2>&1; tail -1 gpurun_out/r4/dictionary_fuzz_fa_min5.log; grep -c "filter stride" gpurun_out/r4/dictionary_fuzz_fa_min5.log
FUZZ_MIN_LEN=7 NEEDLE_PREFILTER=2 NEEDLE_PAIR_MAX_BYTES=0 python scripts/dictionary_fuzz.py 9500 ${2:-24} > gpurun_out/r4/dictionary_fuzz_fa_min7_level2.log 2>&1; tail -1 gpurun_out/r4/dictionary_fuzz_fa_min7_level2.log; grep -c "filter stride" gpurun_out/r4/dictionary_fuzz_fa_min7_level2.log
FUZZ_MIN_LEN=5 NEEDLE_PREFILTER=2 NEEDLE_PAIR_MAX_BYTES=0 python scripts/dictionary_fuzz.py 9700 ${3:-32} > gpurun_out/r4/dictionary_fuzz_fa_min5_level2.log 2>&1; tail -1 gpurun_out/r4/dictionary_fuzz_fa_min5_level2.log; grep -c "filter stride" gpurun_out/r4/dictionary_fuzz_fa_min5_level2.log
python scripts/dictionary_fuzz.py 9900 ${4:-24} > gpurun_out/r4/dictionary_fuzz_fa_plain.log 2>&1; tail -1 gpurun_out/r4/dictionary_fuzz_fa_plain.log
grep -h "FAILED" gpurun_out/r4/dictionary_fuzz_fa_*.log | head -5 | cut -c1-600
