#!/bin/bash
cd "$GRAFT_REPO_ROOT"
b() { timeout 300 python bench.py --workload $1 --steps 30 --also none --no-cpu-baseline --no-extras $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['achieved']), round(d['roofline'].get('frac'),4))"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "not hbm and not c3" 2>&1 | tail -2
for rep in 1 2; do
  NEEDLE_PACK_WAVES=16 b c2 find_16x64 "--op find"
  b c2 find_15x128 "--op find"
  b c2 contained
done
