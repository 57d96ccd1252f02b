#!/bin/bash
# round 4: where the n-gram filter kernel's time goes on the C3-sparse batch (tuning build, scripts/build_tuning.sh):
# NEEDLE_NG_DBG 0 = the product's kernel, 1 = candidates dropped (filter + queue only), 2 = text gathered, no walk, 3 = walk on zeros
# (no gather), 16 = runs of 32 candidates.  Answers differ from the product's in 1..3: timing only.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
NEEDLE_PREFILTER=1 timeout 300 python scripts/r4_ngram.py 2>&1 | grep -v amdgpu | tail -1
for d in 0 1 2 3 16; do echo "== NEEDLE_NG_DBG=$d"; NEEDLE_LIB=$PWD/needle_amd/libneedle_hip_tuning.so NEEDLE_NG_DBG=$d timeout 300 python scripts/r4_ngram.py 2>&1 | grep -v amdgpu | tail -1; done
