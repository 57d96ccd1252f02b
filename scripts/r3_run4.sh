#!/bin/bash
cd "$GRAFT_REPO_ROOT"
scripts/gpu_tests.sh
timeout 1500 python scripts/fuzz_campaign.py 5000 200 > gpurun_out/r3_fuzz.log 2>&1; tail -1 gpurun_out/r3_fuzz.log
python scripts/find_all_probe.py c3 2>/dev/null | tail -1
