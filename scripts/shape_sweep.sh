#!/bin/bash
cd "$GRAFT_REPO_ROOT"
W=$1; shift
for sh in default "$@"; do
  if [ "$sh" = default ]; then unset NEEDLE_SHAPE; else export NEEDLE_SHAPE=$sh; fi
  timeout 300 python bench.py --workload $W --steps 30 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', '$sh', round(d['roofline']['kernel_ms'],4), round(d['roofline']['achieved']))"
done
