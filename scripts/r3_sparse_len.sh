#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_survivor_pool.py tests/test_gpu_configs.py tests/test_gpu_find_forms.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20
b() { timeout 300 python bench.py --workload $1 --steps 30 --also none --no-cpu-baseline --no-extras $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline'].get('frac'),4))"; }
for rep in 1 2; do
  b c3s lengths ""
  NEEDLE_FIND_LENGTHS_SPARSE=0 b c3s backward ""
done
python scripts/r3_dense_dictionary.py 2>&1 | grep -v amdgpu | tail -6
