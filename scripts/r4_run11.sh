#!/bin/bash
# round 4: find-all behind the filter incl. counting / CSR forms: parity, full size, the find-all suites with the filter forced off too
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_prefilter.py tests/test_gpu_full_size.py tests/test_gpu_find_all.py -x -q -m gpu -k "prefilter or c3s or dictionary" > gpurun_out/r4/tests11.log 2>&1; grep -E "passed|failed" gpurun_out/r4/tests11.log | tail -2; grep -E "^E  " gpurun_out/r4/tests11.log | head -12
NEEDLE_PREFILTER=2 NEEDLE_PAIR_MAX_BYTES=0 python -m pytest tests/test_gpu_find_all.py -x -q -m gpu -k "tile_boundaries" > gpurun_out/r4/tests11b.log 2>&1; grep -E "passed|failed" gpurun_out/r4/tests11b.log | tail -2
for e in "NEEDLE_FIND_ALL_FILTER=1" "NEEDLE_FIND_ALL_FILTER=0"; do echo "== $e"; env $e python scripts/find_all_probe.py c3s 10000000 32 check 2>&1 | grep -v amdgpu | tail -1; done | tee gpurun_out/r4/find_all_filter_ab.log
