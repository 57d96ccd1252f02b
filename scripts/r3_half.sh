#!/bin/bash
# round 3: survivor pool -- deferral after the FIRST half of a line (NEEDLE_DEFER_HALF), parity then A/B on C3, same box
cd "$GRAFT_REPO_ROOT"
for h in 24 40; do
  NEEDLE_DEFER_HALF=$h timeout 1500 python -m pytest tests/test_gpu_survivor_pool.py tests/test_gpu_configs.py -x -q -k "16 or c3 or keyword or hbm" 2>&1 | tail -2
done
b() { timeout 300 python bench.py --workload $1 --steps 30 --also none --no-cpu-baseline --no-extras $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['achieved']), round(d['roofline'].get('frac'),4))"; }
for rep in 1 2; do
  for h in 0 12 16 20 24 32 40; do NEEDLE_DEFER_HALF=$h b c3 half$h; done
  for h in 0 24; do NEEDLE_DEFER_HALF=$h b c3 contained_half$h "--op contained_in"; done
  for h in 0 24; do NEEDLE_DEFER_HALF=$h b c3s c3s_half$h; done
done
NEEDLE_DEFER_HALF=0 python scripts/quick_ragged_keywords.py 2>/dev/null | tail -4
NEEDLE_DEFER_HALF=24 python scripts/quick_ragged_keywords.py 2>/dev/null | tail -4
