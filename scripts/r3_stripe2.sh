#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_long_rows.py tests/test_gpu_real_text.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20
for rep in 1 2; do for l in libneedle_hip.so libneedle_hip_prev.so; do echo "== $l"; NEEDLE_LIB=$PWD/needle_amd/$l python scripts/long_rows_rate.py 1000 1 2>&1 | grep -v amdgpu; NEEDLE_LIB=$PWD/needle_amd/$l python scripts/long_rows_rate.py 4 256 2>&1 | grep -v amdgpu;  done; done
