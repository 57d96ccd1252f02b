#!/bin/bash
cd "$GRAFT_REPO_ROOT"
q() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'matched', round(d['matched_fraction'],4))"; }
B="--also none --no-cpu-baseline --no-extras --graph off"
timeout 300 python bench.py --workload c5 $B 2>/dev/null | q "c5 find"
NEEDLE_DEBUG_NO_BACKWARD=1 timeout 300 python bench.py --workload c5 $B 2>/dev/null | q "c5 find no-backward"
timeout 300 python bench.py --workload c5 --op contained_in $B 2>/dev/null | q "c5 contained_in"
timeout 300 python bench.py --workload c5 --op matches $B 2>/dev/null | q "c5 matches"
timeout 300 python bench.py --workload c2 --op find $B 2>/dev/null | q "c2 find"
for sh in 16x128 12x128 16x64 8x128; do NEEDLE_SHAPE=$sh timeout 300 python bench.py --workload c5 $B 2>/dev/null | q "c5 find shape $sh"; done
scripts/pmc.sh c5 r2a "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" 2>&1 | tail -22
