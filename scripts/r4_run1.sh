#!/bin/bash
# round 4, first GPU pass on the n-gram filter build: -m gpu suite, C3-sparse A/B (NEEDLE_PREFILTER=0/1), default bench line, c3s profile
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
scripts/gpu_tests.sh > gpurun_out/r4/gpu_tests.log 2>&1; tail -3 gpurun_out/r4/gpu_tests.log
for k in 0 1; do NEEDLE_PREFILTER=$k timeout 300 python scripts/r4_ngram.py 2>&1 | grep -v amdgpu; done | tee gpurun_out/r4/ngram_ab.log
timeout 1200 python bench.py > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err; tail -2 gpurun_out/r4/bench_default.err
scripts/profile.sh c3s > gpurun_out/profile_c3s.log 2>&1
G1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM"
G2="SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
G3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
scripts/pmc.sh c3s r4ngram "$G1" "$G2" "$G3" > gpurun_out/pmc_c3s_r4ngram.log 2>&1
tail -15 gpurun_out/pmc_c3s_r4ngram.log
