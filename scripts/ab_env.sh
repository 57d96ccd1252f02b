#!/bin/bash
# A/B of ENVIRONMENT settings inside ONE gpurun call (same box, same build): scripts/ab_env.sh "VAR=a VAR=b ..." workloads...
# (each setting is one word: NEEDLE_PREFILTER_STRIDE=2; "-" = nothing set)
cd "$GRAFT_REPO_ROOT"
VS=$1; shift
mkdir -p gpurun_out/r6
for w in "$@"; do
  for rep in 1 2; do
    for v in $VS; do
      e=""; [ "$v" != "-" ] && e="$v"
      env $e timeout 300 python bench.py --workload $w --steps 30 --also none --no-cpu-baseline --no-extras --full-line 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '$v', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['achieved']), d['roofline'].get('kernel'))" | tee -a gpurun_out/r6/ab_env.log
    done
  done
done
