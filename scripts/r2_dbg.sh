#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_survivor_pool.py -m gpu -x -q -k "3-5-2-16" 2>&1 | grep -E "Error|assert|find|contained|matches" | tail -12
