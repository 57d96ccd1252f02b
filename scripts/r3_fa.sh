#!/bin/bash
# round 3, item 2: find-all with per-state match lengths against the backward-walk form, same box
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_find_all.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -x -q -k "find_all or every_match or fuzz or csr or no_slots" 2>&1 | tail -5
for rep in 1 2; do
  for w in c3 c3s; do
    NEEDLE_FIND_ALL_LENGTHS=0 python scripts/find_all_probe.py $w 2>/dev/null | tail -1
    python scripts/find_all_probe.py $w 2>/dev/null | tail -1
  done
done
