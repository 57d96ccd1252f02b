#!/bin/bash
cd "$GRAFT_REPO_ROOT"
scripts/gpu_tests.sh
mkdir -p gpurun_out/r3
timeout 1200 python bench.py > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err
tail -3 gpurun_out/r3/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3/bench_default.json").read().strip().splitlines()[-1])
def line(n, x):
    hl = x.get("host_landed", {})
    print(n, "ms/step %.4f kernel %.4f frac %.3f | cold %.4f (%.3f) steady %.4f x%d | landed %.3f compact %s packed16 %s | %s" % (
        x["ms_per_step"], x["roofline"]["kernel_ms"], x["roofline"]["frac"], x["cold"]["ms_per_step"], x["cold"]["roofline.frac"], x["steady"]["ms_per_step"], x["steady"]["steps_effective"],
        hl.get("ms_per_step", -1), hl.get("compact", {}).get("ms_per_step"), hl.get("packed16", {}).get("ms_per_step"), x["config"]["automaton"]["kernel_mode"]))
line("c2", d)
for k, v in d["workloads"].items():
    if "error" in v: print(k, v)
    else: line(k, v)
print(d.get("c4_shard_step"))
PY
