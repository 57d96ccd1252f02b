#!/bin/bash
# usage: scripts/pmc.sh <workload> <tag> "<COUNTER ...>" ["<COUNTER ...>" ...]   (one rocprofv3 --pmc pass per quoted group)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
W=$1; TAG=$2; shift 2
OUT=gpurun_out/pmc_${W}_${TAG}
mkdir -p $OUT
i=0
for grp in "$@"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python bench.py --workload $W $NEEDLE_BENCH_EXTRA --steps 3 --warmup 1 --also none --no-cpu-baseline --no-extras --full-line > $OUT/p$i.json 2> $OUT/p$i.log
  # only the scan kernel's counter rows are kept: gpurun copies back at most 64 MiB
  python - "$OUT/p$i/p_counter_collection.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if ("scan_kernel" in r["Kernel_Name"] or "ngram_kernel" in r["Kernel_Name"])]
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
    for r in csv.DictReader(open(f)):
        if ("scan_kernel" in r["Kernel_Name"] or "ngram_kernel" in r["Kernel_Name"]):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print("kernel dur us (under pmc):", sum(dur)/len(dur)/1e3)
for k, v in agg.items():
    print("%-28s %.4g" % (k, sum(v)/len(v)))
PY
