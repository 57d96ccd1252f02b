#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fa_packed; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_find_all.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20
for rep in 1 2; do
for w in c3 c3s c5; do
  FIND_ALL_PROBE_DENSE_ONLY=1 python scripts/find_all_probe.py $w 10000000 32 check 2>/dev/null | tail -1
  FIND_ALL_PROBE_DENSE_ONLY=1 FIND_ALL_PROBE_PACKED=1 python scripts/find_all_probe.py $w 10000000 32 check 2>/dev/null | tail -1
done; done
for c in FETCH_SIZE WRITE_SIZE; do
FIND_ALL_PROBE_PACKED=1 FIND_ALL_PROBE_DENSE_ONLY=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o p -- python scripts/find_all_probe.py c3 10000000 32 > /dev/null 2>&1
python - "$O/$c/p_counter_collection.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "find_all" in r["Kernel_Name"]]
print(rows[0]["Counter_Name"], sum(float(r["Counter_Value"]) for r in rows) / len(rows), "KiB per launch,", len(rows), "launches")
PY
rm -rf $O/$c
done
