#!/bin/bash
# round 5: ONE parameterised gpurun wrapper (replaces the r3_* / r4_run* one-shots): scripts/r5.sh <step> [args...]
#   tests <pytest args>     pytest -m gpu on the given files / -k expression -> gpurun_out/r5/tests_<tag>.log
#   fa [workload] [rows]    find-all probe: lock-step kernel vs the one-pass kernel, both result forms (A/B in one call)
#   fa_prof                 rocprofv3 --kernel-trace --stats + PMC passes of the find-all kernels on C3 -> gpurun_out/r5/fa_prof/
#   bench [args]            python bench.py [args] -> gpurun_out/r5/bench_<tag>.json
#   py <script> [args]      python <script> [args] > gpurun_out/r5/<basename>.log
#   pmc_py <tag> <script> [args]   PMC passes (VALU / SALU / LDS instructions, LDS conflicts, waits) of any script: per kernel, per launch
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5; mkdir -p $O
step=$1; shift
TAG=${R5_TAG:-$step}
case $step in
tests)
  python -m pytest "$@" -x -q -m gpu > $O/tests_$TAG.log 2>&1
  grep -E "passed|failed|error" $O/tests_$TAG.log | tail -3; grep -E "^E  " $O/tests_$TAG.log | head -12 ;;
fa)
  w=${1:-c3}; n=${2:-10000000}
  for ls in 1 0; do for pk in "" 1; do
    NEEDLE_FIND_ALL_LOCKSTEP=$ls FIND_ALL_PROBE_PACKED=$pk python scripts/find_all_probe.py $w $n 32 check 2>$O/fa_err.log | tail -1 > $O/fa_${w}_ls${ls}_pk${pk:-0}.json
    echo "lockstep=$ls packed=${pk:-0}: $(cat $O/fa_${w}_ls${ls}_pk${pk:-0}.json)"
  done; done ;;
fa_prof)
  P=$O/fa_prof; mkdir -p $P
  for w in ${@:-c3}; do
    FIND_ALL_PROBE_PACKED=1 FIND_ALL_PROBE_DENSE_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $P/${w}_packed -o t -- python scripts/find_all_probe.py $w 10000000 32 > $P/${w}_packed.json 2> $P/err.log
    FIND_ALL_PROBE_DENSE_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$w -o t -- python scripts/find_all_probe.py $w 10000000 32 > $P/$w.json 2>> $P/err.log
    for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
      t=$(echo $c | tr ' ' '_')
      FIND_ALL_PROBE_PACKED=1 FIND_ALL_PROBE_DENSE_ONLY=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/pmc_$t -o p -- python scripts/find_all_probe.py $w 10000000 32 > /dev/null 2>> $P/err.log
      python - "$P/pmc_$t/p_counter_collection.csv" <<'PY' >> $P/${w}_pmc.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "find_all" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
calls = collections.Counter((r["Counter_Name"]) for r in rows)
k0 = rows[0]["Kernel_Name"][:60] if rows else "?"
for k, v in agg.items():
    # (one row per dispatch and counter dimension: sum per dispatch = total / dispatches)
    n_disp = len(set(r["Dispatch_Id"] for r in rows))
    print("%-26s %.5g   per launch (%d launches, %s)" % (k, sum(v) / max(1, n_disp), n_disp, k0))
PY
      rm -rf $P/pmc_$t
    done
    find $P -name "*_kernel_trace.csv" -delete; find $P -name "*_agent_info.csv" -delete
    head -5 $P/$w/t_kernel_stats.csv | cut -c1-160; head -5 $P/${w}_packed/t_kernel_stats.csv | cut -c1-160; cat $P/${w}_pmc.txt
  done ;;
bench)
  timeout 1700 python bench.py "$@" > $O/bench_$TAG.json 2> $O/bench_$TAG.err; grep -v "^{" $O/bench_$TAG.err | tail -3; wc -c $O/bench_$TAG.json; tail -c 1200 $O/bench_$TAG.json ;;
profiles)
  # everything profiles/r05_* is made from (then: summarize_profile.py gpurun_out/prof_<w> 05 <w>, summarize_pmc.py 05)
  for w in c2 c3 c3s c3x c5 c5w c3x16; do scripts/profile.sh $w > gpurun_out/profile_$w.log 2>&1; done
  G1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM"
  G2="SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
  G3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
  scripts/pmc.sh c3s r5final "$G1" "$G2" "$G3" > gpurun_out/pmc_c3s_r5final.log 2>&1
  NEEDLE_PREFILTER_LEVEL2=0 scripts/pmc.sh c3s r5nolevel2 "$G1" "$G2" "$G3" > gpurun_out/pmc_c3s_r5nolevel2.log 2>&1
  scripts/pmc.sh c3x r5 "$G1" "$G2" "$G3" > gpurun_out/pmc_c3x_r5.log 2>&1
  NEEDLE_PREFILTER=0 scripts/pmc.sh c3x r5scan "$G1" "$G2" "$G3" > gpurun_out/pmc_c3x_r5scan.log 2>&1
  scripts/pmc.sh c3 r5 "$G1" "$G2" "$G3" > gpurun_out/pmc_c3_r5.log 2>&1
  scripts/pmc.sh c5 r5 "$G1" "$G2" "$G3" > gpurun_out/pmc_c5_r5.log 2>&1
  scripts/pmc.sh c5w r5 "$G1" "$G2" "$G3" > gpurun_out/pmc_c5w_r5.log 2>&1
  rm -rf $O/fa_prof; $0 fa_prof c3 > $O/fa_prof.log 2>&1
  NEEDLE_FIND_ALL_LOCKSTEP=0 R5_SUB=ls0 bash -c 'P=gpurun_out/r5/fa_prof_ls0; mkdir -p $P; FIND_ALL_PROBE_PACKED=1 FIND_ALL_PROBE_DENSE_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $P/c3_packed -o t -- python scripts/find_all_probe.py c3 10000000 32 > $P/c3_packed.json 2> $P/err.log; find $P -name "*_kernel_trace.csv" -delete; find $P -name "*_agent_info.csv" -delete'
  for w in c3 c3s c3x c2 c5; do python scripts/find_all_probe.py $w 10000000 32 check 2>/dev/null | tail -1 > $O/fa_$w.json; FIND_ALL_PROBE_PACKED=1 python scripts/find_all_probe.py $w 10000000 32 2>/dev/null | tail -1 > $O/fa_${w}_packed.json; done
  timeout 1700 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python scripts/bench_digest.py $O/bench_default.json ;;
repeat)
  # repeat <workload> <n> [bench args]: the same bench line n times in one lease (timed / cold / steady ms, kernel ms), clocks before and after
  w=$1; n=$2; shift 2
  rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -6
  for i in $(seq $n); do
    python bench.py --workload $w --also none --no-cpu-baseline --no-extras --full-line "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', 'timed', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'cold', round(d['cold']['ms_per_step'],4), 'steady', round(d['steady']['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4))"
  done
  rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -6 ;;
pmc_py)
  tag=$1; shift
  P=$O/pmc_$tag; mkdir -p $P; : > $P.txt
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
    t=$(echo $c | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/$t -o p -- python "$@" > /dev/null 2>> $P/err.log
    python - "$P/$t/p_counter_collection.csv" <<'PY' >> $P.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "needle::" in r["Kernel_Name"]]
agg = collections.defaultdict(float); disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for (k, c), v in sorted(agg.items()):
    print("%-72s %-24s %.5g per launch (%d launches)" % (k, c, v / len(disp[k]), len(disp[k])))
PY
    rm -rf $P/$t
  done
  cat $P.txt ;;
py)
  s=$1; shift
  python $s "$@" > $O/$(basename $s .py)_$TAG.log 2>&1; tail -40 $O/$(basename $s .py)_$TAG.log ;;
*) echo "unknown step $step"; exit 2 ;;
esac
