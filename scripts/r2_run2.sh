#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2b; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2b/bench_default.json").read().strip().splitlines()[-1])
def line(n, x): print(n, "ms/step %.4f kernel %.4f frac %.3f host-issue %.1f us landed %.3f ms cpu %.1f GB/s (1 core %.2f) matched %.4f mode %s" % (x["ms_per_step"], x["roofline"]["kernel_ms"], x["roofline"]["frac"], x["host_issue_us_per_step"], x["host_landed"]["ms_per_step"], x["cpu_baseline"]["value"], x["cpu_baseline"]["single_core"]["value"], x["matched_fraction"], x["config"]["automaton"]))
line("c2", d)
for k, v in d["workloads"].items(): line(k, v)
PY
q() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'issue us', round(d['host_issue_us_per_step'],1), d['config']['launch'], d.get('scan_ms'), d.get('gather_ms'))"; }
for w in c2 c3; do
  for g in off scan; do
    timeout 300 python bench.py --workload $w --rows 1250000 --steps 200 --warmup 20 --graph $g --also none --no-cpu-baseline --no-extras 2>/dev/null | q "$w 1.25M graph=$g"
  done
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload $w --rows 1250000 --steps 200 --warmup 20 --also none --no-cpu-baseline --no-extras 2>$O/dist_$w.err | q "$w 1.25M dist(1 rank)"
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --workload $w --steps 20 --also none --no-cpu-baseline --no-extras 2>>$O/dist_$w.err | q "$w 10M dist(1 rank)"
done
NEEDLE_DEBUG_NO_BACKWARD=1 timeout 300 python bench.py --workload c3 --also none --no-cpu-baseline --no-extras 2>/dev/null | q "c3 no-backward"
timeout 300 python bench.py --workload c3 --op contained_in --also none --no-cpu-baseline --no-extras 2>/dev/null | q "c3 contained_in"
