#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3
for rep in 1 2; do
NEEDLE_SPARSE=0 python scripts/r3_dense_dictionary.py 2>/dev/null | tail -2
python scripts/r3_dense_dictionary.py 2>/dev/null | tail -2
done
python scripts/prefix_prefilter_ab.py 2>gpurun_out/r3/prefix_err.log | tail -8
tail -3 gpurun_out/r3/prefix_err.log
