import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c2", d["ms_per_step"], d["roofline"]["frac"], d.get("cold_ms"), d.get("steady_ms"), d.get("host_landed_ms"))
for k,v in d["workloads"].items():
    print(k, v.get("ms"), v.get("frac"), "cold", v.get("cold_ms"), "steady", v.get("steady_ms"), v.get("host_landed_ms"), v.get("find_all"))
