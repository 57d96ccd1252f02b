#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
q() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'matched', round(d['matched_fraction'],4))"; }
B="--also none --no-cpu-baseline --no-extras --graph off"
for d in 0 8 16 24 32; do NEEDLE_DEFER=$d timeout 300 python bench.py --workload c3 $B 2>/dev/null | q "c3 find defer=$d"; done
for d in 0 24; do NEEDLE_DEFER=$d timeout 300 python bench.py --workload c3 --op contained_in $B 2>/dev/null | q "c3 contained_in defer=$d"; done
for d in 0 24; do NEEDLE_DEFER=$d timeout 300 python bench.py --workload c3 --op matches $B 2>/dev/null | q "c3 matches defer=$d"; done
for w in c2 c5; do timeout 300 python bench.py --workload $w $B 2>/dev/null | q "$w"; done
