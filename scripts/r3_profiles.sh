#!/bin/bash
# everything profiles/r03_* is made from, in one gpurun call (same protocol as bench.py's default line: 150 ms device pre-warm,
# W warm-up steps, K timed steps)
cd "$GRAFT_REPO_ROOT"
scripts/gpu_tests.sh
python scripts/long_find_probe.py Sherlock find 2>/dev/null | tail -1
python scripts/long_find_probe.py Sherlock contained_in 2>/dev/null | tail -1
for w in c2 c3 c3s c5; do scripts/profile.sh $w > gpurun_out/profile_$w.log 2>&1; done
scripts/profile_aux.sh > gpurun_out/profile_aux.log 2>&1
G1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM"
G2="SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
G3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
NEEDLE_SPARSE=0 NEEDLE_WINDOW=0 scripts/pmc.sh c3s r3hybrid "$G1" "$G2" "$G3" > gpurun_out/pmc_c3s_r3hybrid.log 2>&1
NEEDLE_WINDOW=0 NEEDLE_FIND_LENGTHS_SPARSE=0 scripts/pmc.sh c3s r3sparse "$G1" "$G2" "$G3" > gpurun_out/pmc_c3s_r3sparse.log 2>&1
NEEDLE_FIND_LENGTHS_SPARSE=0 scripts/pmc.sh c3s r3sparsewin "$G1" "$G2" "$G3" > gpurun_out/pmc_c3s_r3sparsewin.log 2>&1
scripts/pmc.sh c3s r3sparselen "$G1" "$G2" "$G3" > gpurun_out/pmc_c3s_r3sparselen.log 2>&1
NEEDLE_WINDOW=0 NEEDLE_FIND_LENGTHS=0 scripts/pmc.sh c3 r3cmap "$G1" "$G2" "$G3" > gpurun_out/pmc_c3_r3cmap.log 2>&1
NEEDLE_FIND_LENGTHS=0 scripts/pmc.sh c3 r3window "$G1" "$G2" "$G3" > gpurun_out/pmc_c3_r3window.log 2>&1
scripts/pmc.sh c3 r3lengths "$G1" "$G2" "$G3" > gpurun_out/pmc_c3_r3lengths.log 2>&1
scripts/pmc.sh c5 r3 "$G1" "$G2" "$G3" > gpurun_out/pmc_c5_r3.log 2>&1
scripts/profile_find_all.sh > gpurun_out/profile_find_all.log 2>&1
mkdir -p gpurun_out/r3
for k in 0 1; do echo "== NEEDLE_FIND_LENGTHS=$k"; NEEDLE_FIND_LENGTHS=$k timeout 300 python scripts/find_forms_ab.py 2>&1 | grep -v amdgpu; done > gpurun_out/r3/find_forms_ab.log
timeout 1200 python bench.py > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err
tail -2 gpurun_out/r3/bench_default.err
