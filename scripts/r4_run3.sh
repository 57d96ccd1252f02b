#!/bin/bash
# round 4: c5w (wide C5), the 8-rank rehearsal on one GPU, packed find -- tests, then the default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_configs.py tests/test_gpu_full_size.py tests/test_gpu_multi_device.py tests/test_gpu_find_packed16.py -x -q -m gpu -k "c5w or rehearsal or multi or packed16" > gpurun_out/r4/tests3.log 2>&1; grep -E "passed|failed|error" gpurun_out/r4/tests3.log | tail -3; grep -E "^E  |Error" gpurun_out/r4/tests3.log | head -20
timeout 1500 python bench.py > gpurun_out/r4/bench_default2.json 2> gpurun_out/r4/bench_default2.err; tail -2 gpurun_out/r4/bench_default2.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/bench_default2.json").read().strip().splitlines()[-1])
print("c2", d["ms_per_step"], d["roofline"]["frac"])
for k,v in d.get("workloads",{}).items(): print(k, v.get("ms_per_step"), v.get("roofline",{}).get("frac"), v.get("roofline",{}).get("kernel"), v.get("error"))
print("host_landed", {k:(v if not isinstance(v,dict) else v.get("ms_per_step")) for k,v in d.get("host_landed",{}).items()})
print("c4_shard_step", d.get("c4_shard_step"))
PY
