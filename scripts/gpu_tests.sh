#!/bin/bash
# the whole -m gpu suite on the GPU box; summary line last (RCCL's banner otherwise hides it)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q "$@" > gpurun_out/gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -5
