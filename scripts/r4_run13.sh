#!/bin/bash
# round 4: the default bench line with c4_shard_step in the one-dword result form, the multi-device tests
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_multi_device.py -x -q -m gpu > gpurun_out/r4/tests13.log 2>&1; grep -E "passed|failed" gpurun_out/r4/tests13.log | tail -2; grep -E "^E  " gpurun_out/r4/tests13.log | head -6
timeout 1500 python bench.py > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err; tail -1 gpurun_out/r4/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/bench_default.json").read().strip().splitlines()[-1])
print("c2", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"].get("traffic_source"))
for k,v in d["workloads"].items(): print(k, v["ms_per_step"], v["roofline"]["frac"], v["roofline"]["traffic"])
print(d["c4_shard_step"])
PY
