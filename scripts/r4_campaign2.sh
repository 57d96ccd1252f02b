#!/bin/bash
# round 4: random dictionaries that DO get an n-gram filter (keywords of >= 5 / >= 7 chars: strides 2 and 4), near misses planted:
# default level (compressed automata) and NEEDLE_PREFILTER=2 (every LDS-table automaton)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
FUZZ_MIN_LEN=5 python scripts/dictionary_fuzz.py 3000 ${1:-48} > gpurun_out/r4/dictionary_fuzz_min5.log 2>&1; tail -1 gpurun_out/r4/dictionary_fuzz_min5.log; grep -c "filter stride" gpurun_out/r4/dictionary_fuzz_min5.log
FUZZ_MIN_LEN=7 python scripts/dictionary_fuzz.py 4000 ${2:-24} > gpurun_out/r4/dictionary_fuzz_min7.log 2>&1; tail -1 gpurun_out/r4/dictionary_fuzz_min7.log; grep -c "filter stride" gpurun_out/r4/dictionary_fuzz_min7.log
FUZZ_MIN_LEN=5 NEEDLE_PREFILTER=2 NEEDLE_PAIR_MAX_BYTES=0 python scripts/dictionary_fuzz.py 5000 ${3:-48} > gpurun_out/r4/dictionary_fuzz_min5_level2.log 2>&1; tail -1 gpurun_out/r4/dictionary_fuzz_min5_level2.log; grep -c "filter stride" gpurun_out/r4/dictionary_fuzz_min5_level2.log
FUZZ_MIN_LEN=7 NEEDLE_PREFILTER=2 NEEDLE_PAIR_MAX_BYTES=0 python scripts/dictionary_fuzz.py 6000 ${4:-24} > gpurun_out/r4/dictionary_fuzz_min7_level2.log 2>&1; tail -1 gpurun_out/r4/dictionary_fuzz_min7_level2.log; grep -c "filter stride" gpurun_out/r4/dictionary_fuzz_min7_level2.log
grep -h "filter stride" gpurun_out/r4/dictionary_fuzz_min*.log | sed 's/.*n-gram filter \(stride [0-9]* run-up [0-9]*\).*/\1/' | sort | uniq -c
