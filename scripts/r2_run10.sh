#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
mkdir -p gpurun_out/r2c
timeout 900 python bench.py > gpurun_out/r2c/bench_default.json 2> gpurun_out/r2c/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2c/bench_default.json").read().strip().splitlines()[-1])
def line(n, x): print(n, "ms/step %.4f kernel %.4f frac %.3f landed %.3f ms cpu %.1f GB/s (1 core %.2f) matched %.4f %s" % (x["ms_per_step"], x["roofline"]["kernel_ms"], x["roofline"]["frac"], x["host_landed"]["ms_per_step"], x["cpu_baseline"]["value"], x["cpu_baseline"]["single_core"]["value"], x["matched_fraction"], x["config"]["automaton"]["kernel_mode"]))
line("c2", d)
for k, v in d["workloads"].items(): line(k, v)
PY
