#!/bin/bash
# rocprofv3 evidence for the bench line (run on the GPU box through gpurun): kernel-trace stats, then the HBM
# traffic counters in their own passes (never combined with other trace domains).
set -x
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
W=${1:-c2}
OUT=gpurun_out/prof_$W
mkdir -p $OUT
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os;print(len(os.sched_getaffinity(0)))"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --workload $W --steps 20 --warmup 3 --also none --no-cpu-baseline --no-extras --full-line > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python bench.py --workload $W --steps 5 --warmup 1 --also none --no-cpu-baseline --no-extras --full-line > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python bench.py --workload $W --steps 5 --warmup 1 --also none --no-cpu-baseline --no-extras --full-line > $OUT/bench_write.json 2> $OUT/write.log
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/ea -o e -- python bench.py --workload $W --steps 5 --warmup 1 --also none --no-cpu-baseline --no-extras --full-line > $OUT/bench_ea.json 2> $OUT/ea.log
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/eaw -o e -- python bench.py --workload $W --steps 5 --warmup 1 --also none --no-cpu-baseline --no-extras --full-line > $OUT/bench_eaw.json 2> $OUT/eaw.log
# keep what summarize_profile.py reads (gpurun copies back at most 64 MiB): the stats table and the scan kernel's counter rows
python - $OUT <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "*", "*_counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if ("scan_kernel" in r["Kernel_Name"] or "ngram_kernel" in r["Kernel_Name"])]
    if rows:
        w = csv.DictWriter(open(f, "w", newline=""), fieldnames=list(rows[0].keys()))
        w.writeheader(); w.writerows(rows)
for pat in ("*_kernel_trace.csv", "*_agent_info.csv", "*_domain_stats.csv"):
    for f in glob.glob(os.path.join(sys.argv[1], "*", pat)):
        os.remove(f)
PY
du -sh $OUT
