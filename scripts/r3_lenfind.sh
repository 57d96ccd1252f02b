#!/bin/bash
# find() by the "lengths" automaton (no backward walk) vs the ordinary program: parity first, then the A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -5
grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/gpu_tests.log | head -20
timeout 900 python scripts/fuzz_campaign.py 5000 120 2>&1 | tail -6
b() { timeout 300 python bench.py --workload $1 --steps 30 --also none --no-cpu-baseline --no-extras $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline'].get('frac'),4))"; }
for rep in 1 2; do
  for w in c3 c5 c2; do
    x=""; [ $w = c2 ] && x="--op find"
    b $w lengths "$x"
    NEEDLE_FIND_LENGTHS=0 b $w backward "$x"
  done
done
for k in 1 0; do echo "ragged NEEDLE_FIND_LENGTHS=$k"; NEEDLE_FIND_LENGTHS=$k timeout 300 python scripts/quick_ragged_keywords.py 2>&1 | grep -v amdgpu | tail -6; done
for k in 1 0; do echo "short NEEDLE_FIND_LENGTHS=$k"; NEEDLE_FIND_LENGTHS=$k timeout 300 python scripts/short_rows_rate.py 32 "$(python -c "import sys; sys.path.insert(0, \".\"); from needle_amd import workload as W; print(\"|\".join(W.keywords(300)))")" 2>&1 | grep -v amdgpu | tail -8; done
