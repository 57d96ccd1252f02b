#!/bin/bash
# round 4: differential campaigns on the final build (n-gram filter on by default): random dictionaries (default level and
# NEEDLE_PREFILTER=2 with plain tables), random regexes, the dense-dictionary and decapitated-keyword texts with the filter off / on
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python scripts/dictionary_fuzz.py 1000 ${1:-96} > gpurun_out/r4/dictionary_fuzz.log 2>&1; tail -1 gpurun_out/r4/dictionary_fuzz.log
NEEDLE_PREFILTER=2 NEEDLE_PAIR_MAX_BYTES=0 python scripts/dictionary_fuzz.py 2000 ${2:-32} > gpurun_out/r4/dictionary_fuzz_level2.log 2>&1; tail -1 gpurun_out/r4/dictionary_fuzz_level2.log
python scripts/fuzz_campaign.py 5000 ${3:-200} > gpurun_out/r4/fuzz_campaign.log 2>&1; tail -2 gpurun_out/r4/fuzz_campaign.log
for k in 0 1; do NEEDLE_PREFILTER=$k python scripts/r3_dense_dictionary.py 2>&1 | grep -v amdgpu | tail -2; NEEDLE_PREFILTER=$k python scripts/r4_ngram_worstcase.py 2>&1 | grep -v amdgpu | tail -2; done | tee gpurun_out/r4/ngram_hard_texts.log
