#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
q() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'matched', round(d['matched_fraction'],4), d['config']['automaton'])"; }
B="--also none --no-cpu-baseline --no-extras --graph off"
for w in c3 c3s; do for d in 0 12 16 20; do NEEDLE_DEFER=$d timeout 400 python bench.py --workload $w $B 2>/dev/null | q "$w find defer=$d"; done; done
NEEDLE_HYBRID=0 timeout 400 python bench.py --workload c3s $B 2>/dev/null | q "c3s find hybrid=0"
timeout 400 python bench.py --workload c3s --op contained_in $B 2>/dev/null | q "c3s contained_in"
python scripts/quick_ragged.py "$(python -c "
import sys; sys.path.insert(0,'.')
from needle_amd import workload as W; print('|'.join(W.keywords(1000)))")" 2>/dev/null | tail -9
