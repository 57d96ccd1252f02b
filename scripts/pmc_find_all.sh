#!/bin/bash
# PMC passes over the one-pass find-all kernel: scripts/pmc_find_all.sh <workload> <tag> "<COUNTER ...>" ...
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
W=$1; TAG=$2; shift 2
OUT=gpurun_out/pmc_fa_${W}_${TAG}
mkdir -p $OUT
i=0
for grp in "$@"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- env FIND_ALL_PROBE_DENSE_ONLY=1 python scripts/find_all_probe.py $W 10000000 32 > $OUT/p$i.json 2> $OUT/p$i.log
  python - "$OUT/p$i/p_counter_collection.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "find_all" in r["Kernel_Name"]]
if rows:
    w = csv.DictWriter(open(sys.argv[1], "w", newline=""), fieldnames=list(rows[0].keys()))
    w.writeheader(); w.writerows(rows)
PY
  rm -f $OUT/p$i/p_kernel_trace.csv $OUT/p$i/p_agent_info.csv
  i=$((i+1))
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
dur = []
for f in sorted(glob.glob("$OUT/p*/p_counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    names = sorted(set(r["Counter_Name"] for r in rows))
    for k in names:
        mine = [r for r in rows if r["Counter_Name"] == k]
        calls = max(1, sum(1 for r in mine if "find_all_kernel" in r["Kernel_Name"]))
        agg[k].append(sum(float(r["Counter_Value"]) for r in mine) / calls)
        if k == names[0]:
            dur.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in mine) / calls)
print("kernel dur us (under pmc):", sum(dur)/len(dur)/1e3)
for k, v in agg.items():
    print("%-28s %.4g" % (k, sum(v)/len(v)))
PY
