#!/bin/bash
# Round 3, item 1: the compressed automaton (MODE_SPARSE) against the hot-rows mode on the sparse-match dictionary, same box.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3
b() { timeout 300 python bench.py --workload $1 --steps 30 --also none --no-cpu-baseline --no-extras 2>>gpurun_out/r3/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['achieved']), d['roofline'].get('frac'))"; }
if [ "$1" = "test" ]; then timeout 900 python -m pytest tests/test_gpu_survivor_pool.py -x -q -k "6-8" 2>&1 | tail -5; fi
for rep in 1 2; do
  NEEDLE_SPARSE=0 b c3s hybrid
  b c3s sparse
done
