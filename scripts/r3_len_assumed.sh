#!/bin/bash
# upper bound of a lengths-specialised instantiation of the scan kernel (no snapshot / backward code compiled in)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for l in libneedle_hip.so libneedle_hip_tuning_len.so; do echo "== $l"; NEEDLE_LIB=$PWD/needle_amd/$l timeout 300 python scripts/quick_ragged_keywords.py 2>&1 | grep find; 
NEEDLE_LIB=$PWD/needle_amd/$l timeout 300 python scripts/find_forms_ab.py 2>&1 | grep -E "kw40|kw300|kw1000"; done; done
