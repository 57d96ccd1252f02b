"""One line per workload of bench.py's stdout digest: python scripts/bench_digest.py <bench json>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bytes", len(json.dumps(d)))
print("c2", d["ms_per_step"], d["roofline"]["frac"], d.get("cold_ms"), d.get("steady_ms"), d.get("host_ms"), d.get("find_all"))
for k, v in d.get("workloads", {}).items():
    print(k, v.get("ms"), v.get("kernel_ms"), v.get("frac"), "cold", v.get("cold_ms"), "steady", v.get("steady_ms"), v.get("host_ms"), v.get("find_all"), v.get("error"))
print("ragged", d.get("ragged"))
print("c4_shard_step", d.get("c4_shard_step"))
