#!/bin/bash
cd "$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_survivor_pool.py -x -q -k "6-8-6 and 16" 2>&1 | tail -2
b() { timeout 300 python bench.py --workload $1 --steps 30 --also none --no-cpu-baseline --no-extras $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['achieved']), round(d['roofline'].get('frac'),4))"; }
for rep in 1 2; do
  NEEDLE_DICT=0 b c3s tiled
  NEEDLE_DICT=1 b c3s dict
  NEEDLE_DICT=0 b c3s tiled_contained "--op contained_in"
  NEEDLE_DICT=1 b c3s dict_contained "--op contained_in"
done
mkdir -p gpurun_out/pmc_dict
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_dict/f -o p -- python bench.py --workload c3s --steps 3 --warmup 1 --also none --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_dict/s -o p -- python bench.py --workload c3s --steps 3 --warmup 1 --also none --no-cpu-baseline --no-extras > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_dict/*/p_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "dict_kernel" in r["Kernel_Name"] or "backward_row" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(k, "%.4g" % (sum(v)/len(v)))
PY
rm -rf gpurun_out/pmc_dict/*/p_kernel_trace.csv gpurun_out/pmc_dict/*/p_agent_info.csv gpurun_out/pmc_dict/*/p_counter_collection.csv
