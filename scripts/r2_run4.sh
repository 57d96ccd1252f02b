#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
q() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'matched', round(d['matched_fraction'],4))"; }
B="--also none --no-cpu-baseline --no-extras --graph off"
for w in c2 c3 c5; do timeout 300 python bench.py --workload $w $B 2>/dev/null | q "$w"; done
timeout 300 python bench.py --workload c5 --op contained_in $B 2>/dev/null | q "c5 contained_in"
timeout 300 python bench.py --workload c2 --op find $B 2>/dev/null | q "c2 find"
scripts/pmc.sh c5 r2b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" 2>&1 | tail -16
