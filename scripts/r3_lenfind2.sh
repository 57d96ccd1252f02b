#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_find_forms.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20
for k in 0 1 2; do echo "== NEEDLE_FIND_LENGTHS=$k"; NEEDLE_FIND_LENGTHS=$k timeout 300 python scripts/find_forms_ab.py 2>&1 | grep -v amdgpu; done
