#!/bin/bash
# round 4: pending checks in one call -- the 8-rank rehearsal test, short-row find with the one-dword result form, the literal-prefix
# A/B re-taken on the current kernels (profiles/r04_prefilter_ab.json), the find-all probe as a reference for the work on it
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_multi_device.py -x -q -m gpu -k rehearsal -s > gpurun_out/r4/tests4.log 2>&1; grep -E "passed|failed|C4 rehearsal|^E  " gpurun_out/r4/tests4.log | tail -8
for s in 16 32 64; do python scripts/short_rows_rate.py $s 2>&1 | grep -v amdgpu; done | tee gpurun_out/r4/short_rows.log
python scripts/prefix_prefilter_ab.py 2>gpurun_out/r4/prefix_err.log | tail -6
for w in c3 c3s; do python scripts/find_all_probe.py $w 2>&1 | grep -v amdgpu | tail -3; done | tee gpurun_out/r4/find_all_ref.log
G1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM"
G2="SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
G3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
for w in c5w c5; do scripts/pmc.sh $w r4 "$G1" "$G2" "$G3" > gpurun_out/pmc_${w}_r4.log 2>&1; echo "== $w"; tail -14 gpurun_out/pmc_${w}_r4.log; done
NEEDLE_BENCH_EXTRA="--op contained_in" scripts/pmc.sh c5w r4contained "$G1" "$G2" "$G3" > gpurun_out/pmc_c5w_r4contained.log 2>&1; echo "== c5w containedIn"; tail -14 gpurun_out/pmc_c5w_r4contained.log
