#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2a
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python scripts/r2_probe.py > gpurun_out/r2a/probe.log 2>&1
cat gpurun_out/r2a/probe.log | grep -v amdgpu.ids
for w in c2 c3 c5; do timeout 300 python bench.py --workload $w --steps 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['achieved']))"; done
scripts/pmc.sh c3 r2before "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" 2>&1 | tail -25
