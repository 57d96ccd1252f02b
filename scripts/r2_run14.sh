#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
scripts/ab.sh "needle_amd/libneedle_hip_prev.so" c5 c3
