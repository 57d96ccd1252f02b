#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for rx in "sherlock|holmes|watson|irene|adler|john|baker" "(foo|bar)[a-z]{3,5}baz" "http://.+"; do
  for pair in 0 98304; do
    echo "== NEEDLE_PAIR_MAX_BYTES=$pair"; NEEDLE_PAIR_MAX_BYTES=$pair python scripts/quick_regex.py c3 "$rx" 2>&1 | grep -v amdgpu
  done
done
