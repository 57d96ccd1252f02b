#!/bin/bash
cd "$GRAFT_REPO_ROOT"
scripts/ab.sh "needle_amd/libneedle_hip_prev.so" c5 c2
