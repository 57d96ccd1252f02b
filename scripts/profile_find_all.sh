#!/bin/bash
# rocprofv3 of the one-pass find-all kernel on the bench workloads' rows (10M x 256): --kernel-trace --stats per
# workload, the round-per-match form beside it, then the PMC passes on the C3 dictionary.  -> gpurun_out/prof_fa/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_fa; mkdir -p $O
for w in c3 c2 c5 c3s; do
  FIND_ALL_PROBE_DENSE_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -o t -- python scripts/find_all_probe.py $w 10000000 32 > $O/$w.prof.json 2> $O/$w.err
  python scripts/find_all_probe.py $w 10000000 32 2>/dev/null | tail -1 > $O/$w.json
  NEEDLE_FIND_ALL_ROUNDS=1 python scripts/find_all_probe.py $w 10000000 32 2>/dev/null | tail -1 > $O/${w}_rounds.json
done
# one dword per match (needle_find_all_packed16_dev) on the dictionary: time, then the HBM counters
FIND_ALL_PROBE_PACKED=1 FIND_ALL_PROBE_DENSE_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_packed -o t -- python scripts/find_all_probe.py c3 10000000 32 > $O/c3_packed.prof.json 2> $O/c3_packed.err
FIND_ALL_PROBE_PACKED=1 FIND_ALL_PROBE_DENSE_ONLY=1 python scripts/find_all_probe.py c3 10000000 32 check 2>/dev/null | tail -1 > $O/c3_packed.json
for c in FETCH_SIZE WRITE_SIZE; do
FIND_ALL_PROBE_PACKED=1 FIND_ALL_PROBE_DENSE_ONLY=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/packed_$c -o p -- python scripts/find_all_probe.py c3 10000000 32 > /dev/null 2>&1
python - "$O/packed_$c/p_counter_collection.csv" <<'PY' > $O/packed_$c.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "find_all" in r["Kernel_Name"]]
print(rows[0]["Counter_Name"], sum(float(r["Counter_Value"]) for r in rows) / len(rows))
PY
rm -rf $O/packed_$c
done
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*_agent_info.csv" -delete
scripts/pmc_find_all.sh c3 r2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" > $O/pmc_c3.txt
FIND_ALL_PROBE_DENSE_ONLY=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python scripts/find_all_probe.py c3 10000000 32 > /dev/null 2>&1
FIND_ALL_PROBE_DENSE_ONLY=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python scripts/find_all_probe.py c3 10000000 32 > /dev/null 2>&1
for d in fetch write; do
python - "$O/$d/p_counter_collection.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "find_all" in r["Kernel_Name"]]
w = csv.DictWriter(open(sys.argv[1], "w", newline=""), fieldnames=list(rows[0].keys()))
w.writeheader(); w.writerows(rows)
PY
rm -f $O/$d/p_kernel_trace.csv $O/$d/p_agent_info.csv
done
for w in c3 c2 c5 c3s; do tail -1 $O/$w.json; cat $O/${w}_rounds.json; head -4 $O/$w/t_kernel_stats.csv | cut -c1-150; done
cat $O/pmc_c3.txt
