#!/bin/bash
# round 4: the n-gram filter's second hash form (dot2 / and-or / SDWA shift) + needle_find_packed16_dev: parity tests, C3-sparse A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_prefilter.py tests/test_gpu_find_packed16.py tests/test_gpu_compact.py tests/test_gpu_multi_device.py -x -q -m gpu > gpurun_out/r4/tests2.log 2>&1; grep -E "passed|failed|error" gpurun_out/r4/tests2.log | tail -3
for k in 0 1; do NEEDLE_PREFILTER=$k timeout 300 python scripts/r4_ngram.py 2>&1 | grep -v amdgpu; done | tee gpurun_out/r4/ngram_ab2.log
G1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM"
G2="SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
G3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
scripts/pmc.sh c3s r4ngram2 "$G1" "$G2" "$G3" > gpurun_out/pmc_c3s_r4ngram2.log 2>&1
tail -14 gpurun_out/pmc_c3s_r4ngram2.log
