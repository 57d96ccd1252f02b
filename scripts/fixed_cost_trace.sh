#!/bin/bash
# kernel durations (rocprofv3 kernel trace) of find() on small batches: what a launch costs before it streams (LDS staging, first groups)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
for w in "$@"; do
  rm -rf /tmp/fc_$w
  rocprofv3 --kernel-trace --output-format csv -d /tmp/fc_$w -o t -- python scripts/fixed_cost_probe.py $w 4096,65536,262144,1000000,2500000 5 > /tmp/fc_$w.log 2>&1
  f=$(find /tmp/fc_$w -name "*kernel_trace.csv" | head -1)
  python - "$f" $w <<'PY' | tee -a $O/fixed_cost_trace.log
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "needle" not in n: continue
    d.setdefault(n[:70], []).append(round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000, 1))
for k, v in d.items(): print(sys.argv[2], k, v)
PY
done
