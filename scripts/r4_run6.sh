#!/bin/bash
# round 4: the flat page map for UTF-16 table automata (C5w): parity (both maps), A/B on the bench batch
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_find_all.py -x -q -m gpu -k "c5w or fuzz or utf16 or tile_boundaries" > gpurun_out/r4/tests6.log 2>&1; grep -E "passed|failed|^E  " gpurun_out/r4/tests6.log | tail -6
for f in 1 0; do
  for op in find contained_in; do
    NEEDLE_FLAT_MAP=$f timeout 300 python bench.py --workload c5w --op $op --steps 20 --also none --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5w flat=$f $op', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3), d['config']['launch'])"
  done
done | tee gpurun_out/r4/flat_map_ab.log
G1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM"
G2="SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
G3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
scripts/pmc.sh c5w r4flat "$G1" "$G2" "$G3" > gpurun_out/pmc_c5w_r4flat.log 2>&1; tail -14 gpurun_out/pmc_c5w_r4flat.log
