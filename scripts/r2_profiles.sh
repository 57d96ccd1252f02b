#!/bin/bash
# everything profiles/r02_* is made from, in one gpurun call
cd "$GRAFT_REPO_ROOT"
for w in c2 c3 c3s c5; do scripts/profile.sh $w > gpurun_out/profile_$w.log 2>&1; done
scripts/profile_aux.sh > gpurun_out/profile_aux.log 2>&1
scripts/pmc.sh c3 r2after "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" > gpurun_out/pmc_c3_after.log 2>&1
scripts/pmc.sh c3s r2after "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" > gpurun_out/pmc_c3s_after.log 2>&1
scripts/pmc.sh c5 r2after "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" > gpurun_out/pmc_c5_after.log 2>&1
mkdir -p gpurun_out/r2d
timeout 900 python bench.py > gpurun_out/r2d/bench_default.json 2> gpurun_out/r2d/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2d/bench_default.json").read().strip().splitlines()[-1])
def line(n, x): print(n, "ms/step %.4f kernel %.4f frac %.3f traffic %s landed %.3f ms cpu %.1f GB/s (1 core %.2f) matched %.4f %s" % (x["ms_per_step"], x["roofline"]["kernel_ms"], x["roofline"]["frac"], x["roofline"]["traffic"], x["host_landed"]["ms_per_step"], x["cpu_baseline"]["value"], x["cpu_baseline"]["single_core"]["value"], x["matched_fraction"], x["config"]["automaton"]["kernel_mode"]))
line("c2", d)
for k, v in d["workloads"].items(): line(k, v)
PY
