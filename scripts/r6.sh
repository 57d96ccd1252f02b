#!/bin/bash
# round 6: the gpurun wrapper (as scripts/r5.sh): scripts/r6.sh <step> [args...] -> gpurun_out/r6/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O
step=$1; shift
TAG=${R6_TAG:-$step}
case $step in
tests)
  python -m pytest "$@" -x -q -m gpu > $O/tests_$TAG.log 2>&1
  grep -E "passed|failed|error" $O/tests_$TAG.log | tail -3; grep -E "^E  " $O/tests_$TAG.log | head -12 ;;
fa)
  # find-all probe, three result forms (two arrays / one dword row-major / one dword group-blocked), the counting pass and the compact form
  for w in ${@:-c3}; do
    for form in "" PACKED BLOCKED; do
      env ${form:+FIND_ALL_PROBE_$form=1} python scripts/find_all_probe.py $w 10000000 32 check 2>$O/fa_err.log | tail -1 > $O/fa_${w}_${form:-arrays}.json
      echo "$w ${form:-arrays}: $(cat $O/fa_${w}_${form:-arrays}.json)"
    done
  done ;;
fa_prof)
  # rocprofv3 kernel trace + WRITE_SIZE / FETCH_SIZE of the find-all kernel, row-major vs group-blocked one-dword forms
  P=$O/fa_prof; mkdir -p $P
  for w in ${@:-c3}; do for form in PACKED BLOCKED; do
    env FIND_ALL_PROBE_$form=1 FIND_ALL_PROBE_DENSE_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $P/${w}_$form -o t -- python scripts/find_all_probe.py $w 10000000 32 > $P/${w}_$form.json 2> $P/err.log
    for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
      t=$(echo $c | tr ' ' '_')
      env FIND_ALL_PROBE_$form=1 FIND_ALL_PROBE_DENSE_ONLY=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/pmc_$t -o p -- python scripts/find_all_probe.py $w 10000000 32 > /dev/null 2>> $P/err.log
      python scripts/pmc_per_launch.py "$P/pmc_$t/p_counter_collection.csv" find_all >> $P/${w}_${form}_pmc.txt
      rm -rf $P/pmc_$t
    done
    find $P -name "*_kernel_trace.csv" -delete; find $P -name "*_agent_info.csv" -delete
    head -4 $P/${w}_$form/t_kernel_stats.csv | cut -c1-160; cat $P/${w}_${form}_pmc.txt
  done; done ;;
bench)
  timeout 1700 python bench.py "$@" > $O/bench_$TAG.json 2> $O/bench_$TAG.err; grep -v "^{" $O/bench_$TAG.err | tail -3; wc -c $O/bench_$TAG.json; tail -c 1500 $O/bench_$TAG.json ;;
profiles)
  # everything profiles/r06_* is made from: per workload the rocprofv3 kernel trace + FETCH / WRITE / EA passes (scripts/profile.sh), PMC
  # groups of the filter / walk kernels (scripts/pmc.sh), the find-all kernels' trace + counters in both one-dword forms, then the default bench
  for w in ${@:-c2 c3 c3s c3x c5 c5w c3s16 c3x16 c3m16 c3u}; do scripts/profile.sh $w > gpurun_out/profile_$w.log 2>&1; done
  G1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM"
  G2="SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
  G3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
  for w in c3s c3x c5w c3m16; do scripts/pmc.sh $w r6 "$G1" "$G2" "$G3" > gpurun_out/pmc_${w}_r6.log 2>&1; done
  rm -rf $O/fa_prof; $0 fa_prof c3 > $O/fa_prof.log 2>&1
  for w in c3 c2 c5 c3s c3x; do $0 fa $w > $O/fa_$w.log 2>&1; done
  timeout 1700 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python scripts/bench_digest.py $O/bench_default.json ;;
py)
  s=$1; shift
  python $s "$@" > $O/$(basename $s .py)_$TAG.log 2>&1; tail -30 $O/$(basename $s .py)_$TAG.log ;;
sh)
  bash -c "$*" > $O/sh_$TAG.log 2>&1; tail -40 $O/sh_$TAG.log ;;
*) echo "unknown step $step"; exit 2 ;;
esac
