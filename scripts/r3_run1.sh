#!/bin/bash
# round 3, first full check: the whole GPU suite, then PMC passes of the sparse-match dictionary in both big-automaton modes
cd "$GRAFT_REPO_ROOT"
scripts/gpu_tests.sh
G1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM"
G2="SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
G3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
NEEDLE_SPARSE=0 scripts/pmc.sh c3s r3hybrid "$G1" "$G2" "$G3" > gpurun_out/pmc_c3s_r3hybrid.log 2>&1
scripts/pmc.sh c3s r3sparse "$G1" "$G2" "$G3" > gpurun_out/pmc_c3s_r3sparse.log 2>&1
tail -16 gpurun_out/pmc_c3s_r3hybrid.log gpurun_out/pmc_c3s_r3sparse.log
