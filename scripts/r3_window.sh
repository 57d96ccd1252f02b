#!/bin/bash
# round 3: window addressing (no column-map lookup) A/B on the table-mode workloads, same box; parity first
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_survivor_pool.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_long_rows.py tests/test_gpu_real_text.py tests/test_gpu_matches_txt_batch.py -x -q 2>&1 | tail -4
b() { timeout 300 python bench.py --workload $1 --steps 30 --also none --no-cpu-baseline --no-extras $3 2>>gpurun_out/r3/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['achieved']), round(d['roofline'].get('frac'),4), d['config']['automaton']['kernel_mode'])"; }
for rep in 1 2; do
  NEEDLE_WINDOW=0 b c3 cmap
  b c3 window
  NEEDLE_WINDOW=0 b c3s sparse_cmap
  b c3s sparse_window
  NEEDLE_SPARSE=0 NEEDLE_WINDOW=0 b c3s hybrid_cmap
  NEEDLE_SPARSE=0 b c3s hybrid_window
  NEEDLE_WINDOW=0 b c3 cmap_contained "--op contained_in"
  b c3 window_contained "--op contained_in"
done
NEEDLE_WINDOW=0 python scripts/quick_ragged_keywords.py 2>/dev/null | tail -4
python scripts/quick_ragged_keywords.py 2>/dev/null | tail -4
