#!/bin/bash
# upper bound of what find() without any backward walk / snapshot would gain (tuning build: start = end)
cd "$GRAFT_REPO_ROOT"
b() { timeout 300 python bench.py --workload $1 --steps 30 --also none --no-cpu-baseline --no-extras $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline'].get('frac'),4))"; }
export NEEDLE_LIB=$PWD/needle_amd/libneedle_hip_tuning.so
for rep in 1 2; do
  for w in c3 c3s c5 c2; do
    x=""; [ $w = c2 ] && x="--op find"
    b $w with_backward "$x"
    NEEDLE_DEBUG_NO_BACKWARD=1 b $w no_backward "$x"
  done
done
