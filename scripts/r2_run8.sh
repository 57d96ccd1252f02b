#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
q() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'matched', round(d['matched_fraction'],4), d['config']['automaton']['kernel_mode'])"; }
B="--also none --no-cpu-baseline --no-extras --graph off"
for w in c2 c3 c3s c5; do timeout 400 python bench.py --workload $w $B 2>/dev/null | q "$w"; done
python scripts/quick_ragged.py "$(python -c "
import sys; sys.path.insert(0,'.')
from needle_amd import workload as W; print('|'.join(W.keywords(1000)))")" 2>/dev/null | tail -9
scripts/pmc.sh c3s r2a "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" 2>&1 | tail -15
