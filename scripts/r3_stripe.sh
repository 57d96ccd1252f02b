#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_long_rows.py tests/test_gpu_real_text.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20
for rep in 1 2; do for k in 1 0; do echo "NEEDLE_STRIPE_CAND=$k"; NEEDLE_STRIPE_CAND=$k python scripts/long_rows_rate.py 1000 1 2>&1 | grep -v amdgpu; NEEDLE_STRIPE_CAND=$k python scripts/long_rows_rate.py 4 256 2>&1 | grep find; done; done
