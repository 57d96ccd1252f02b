#!/bin/bash
# round 3: the two-sets-per-wave kernel for big automata (needle_dict.hip) -- parity, then A/B against the ordinary tiled kernel
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_survivor_pool.py -x -q -k "6-8-6" 2>&1 | tail -2
NEEDLE_DICT=2 timeout 1500 python -m pytest tests/test_gpu_survivor_pool.py tests/test_gpu_full_size.py -x -q -k "3-5-2 or c3" 2>&1 | tail -2
b() { timeout 300 python bench.py --workload $1 --steps 30 --also none --no-cpu-baseline --no-extras $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['achieved']), round(d['roofline'].get('frac'),4))"; }
for rep in 1 2; do
  NEEDLE_DICT=0 b c3s tiled
  NEEDLE_DICT=1 b c3s dict
  NEEDLE_DICT=0 b c3s tiled_contained "--op contained_in"
  NEEDLE_DICT=1 b c3s dict_contained "--op contained_in"
  NEEDLE_DICT=0 b c3 c3_tiled
  NEEDLE_DICT=2 b c3 c3_dict
  NEEDLE_DICT=0 b c3 c3_tiled_contained "--op contained_in"
  NEEDLE_DICT=2 b c3 c3_dict_contained "--op contained_in"
done
NEEDLE_DICT=0 python scripts/r3_dense_dictionary.py 2>/dev/null | tail -2
NEEDLE_DICT=1 python scripts/r3_dense_dictionary.py 2>/dev/null | tail -2
