#!/bin/bash
# everything profiles/r04_* is made from, in one gpurun call (same protocol as bench.py's default line: 150 ms device pre-warm,
# W warm-up steps, K timed steps); then: summarize_profile.py gpurun_out/prof_<w> 04 <w>, summarize_pmc.py 04, summarize_find_all.py 04
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
scripts/gpu_tests.sh
for w in c2 c3 c3s c5 c5w; do scripts/profile.sh $w > gpurun_out/profile_$w.log 2>&1; done
G1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM"
G2="SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
G3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
NEEDLE_PREFILTER=0 scripts/pmc.sh c3s r4scan "$G1" "$G2" "$G3" > gpurun_out/pmc_c3s_r4scan.log 2>&1
scripts/pmc.sh c3s r4final "$G1" "$G2" "$G3" > gpurun_out/pmc_c3s_r4final.log 2>&1
NEEDLE_BENCH_EXTRA="--op contained_in" scripts/pmc.sh c3s r4finalc "$G1" "$G2" "$G3" > gpurun_out/pmc_c3s_r4finalc.log 2>&1
scripts/pmc.sh c3 r4 "$G1" "$G2" "$G3" > gpurun_out/pmc_c3_r4.log 2>&1
scripts/pmc.sh c5 r4 "$G1" "$G2" "$G3" > gpurun_out/pmc_c5_r4.log 2>&1
NEEDLE_FLAT_MAP=0 scripts/pmc.sh c5w r4 "$G1" "$G2" "$G3" > gpurun_out/pmc_c5w_r4.log 2>&1
scripts/pmc.sh c5w r4flat "$G1" "$G2" "$G3" > gpurun_out/pmc_c5w_r4flat.log 2>&1
NEEDLE_FLAT_MAP=0 NEEDLE_BENCH_EXTRA="--op contained_in" scripts/pmc.sh c5w r4contained "$G1" "$G2" "$G3" > gpurun_out/pmc_c5w_r4contained.log 2>&1
NEEDLE_BENCH_EXTRA="--op contained_in" scripts/pmc.sh c5w r4flatc "$G1" "$G2" "$G3" > gpurun_out/pmc_c5w_r4flatc.log 2>&1
scripts/profile_find_all.sh > gpurun_out/profile_find_all.log 2>&1
timeout 1500 python bench.py > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err
tail -2 gpurun_out/r4/bench_default.err
