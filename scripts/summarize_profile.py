#!/usr/bin/env python3
"""Condenses a scripts/profile.sh output directory (gpurun_out/prof_<w>) into profiles/r<NN>_<w>.json + .md:
kernel-trace stats of needle::scan_kernel and the HBM traffic counters, corrected as MI355X_MICROARCH.md
prescribes (FETCH_SIZE is in KiB and reports exactly 1/2 of a wide coalesced stream on gfx950 -> x2; WRITE_SIZE
in KiB, uncalibrated)."""
import csv
import json
import os
import sys

src, rnd, w = sys.argv[1], sys.argv[2], sys.argv[3]
# FETCH_SIZE -> bytes factor (x 1024 x factor): 2.0 (guide value; re-measured here: a known 2.560 GB read reports
# 1.250e6 KiB).  Cross-check kept beside it: L2 -> fabric read requests, TCC_EA0_RDREQ_{32,64,128}B_sum x their sizes
# (every L2 miss of these kernels is a 128-byte line fetch), and TCC_EA0_WRREQ for the writes.
fetch_factor = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
out = {"workload": w, "round": rnd, "fetch_size_to_bytes_factor": fetch_factor}
for r in csv.DictReader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))):
    if ("scan_kernel" in r["Name"] or "ngram_kernel" in r["Name"]):
        out["kernel"] = r["Name"]
        out["calls"] = int(r["Calls"])
        out["avg_ns"] = float(r["AverageNs"])
        out["min_ns"] = float(r["MinNs"])
        out["max_ns"] = float(r["MaxNs"])
        out["pct_of_gpu_time"] = float(r["Percentage"])
for name, sub, f in (("FETCH_SIZE", "fetch", "f"), ("WRITE_SIZE", "write", "w")):
    vals = []
    for r in csv.DictReader(open(os.path.join(src, sub, f + "_counter_collection.csv"))):
        if ("scan_kernel" in r["Kernel_Name"] or "ngram_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == name:
            vals.append(float(r["Counter_Value"]))
            # (rocprofv3's VGPR_Count / SGPR_Count columns are NOT the code object's .vgpr_count -- r04: 48 here against 91 in the ELF notes of
            # ngram_kernel<2,6,2>: kept under their own names; occupancy reasoning takes scripts/kernel_resources.py's figures)
            out["rocprof_VGPR_Count_column"], out["rocprof_SGPR_Count_column"], out["workgroup"], out["grid"] = r["VGPR_Count"], r["SGPR_Count"], r["Workgroup_Size"], r["Grid_Size"]
    out[name + "_KiB_per_launch"] = sum(vals) / len(vals)
def mean_counters(sub, f):
    agg = {}
    path = os.path.join(src, sub, f + "_counter_collection.csv")
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if ("scan_kernel" in r["Kernel_Name"] or "ngram_kernel" in r["Kernel_Name"]):
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


ea = mean_counters("ea", "e")
eaw = mean_counters("eaw", "e")
if ea:
    r128, r64, r32 = ea.get("TCC_EA0_RDREQ_128B_sum", 0), ea.get("TCC_EA0_RDREQ_64B_sum", 0), ea.get("TCC_EA0_RDREQ_32B_sum", 0)
    other = ea.get("TCC_EA0_RDREQ_sum", 0) - r128 - r64 - r32
    out["ea_read_requests"] = {"128B": r128, "64B": r64, "32B": r32, "unsized": other}
    out["ea_read_bytes_per_launch"] = r128 * 128 + r64 * 64 + r32 * 32
if eaw:
    out["ea_write_requests"] = {"total": eaw.get("TCC_EA0_WRREQ_sum", 0), "64B": eaw.get("TCC_EA0_WRREQ_64B_sum", 0)}
    out["l2_hit"] = eaw.get("TCC_HIT_sum")
    out["l2_miss"] = eaw.get("TCC_MISS_sum")
fetch_b = out["FETCH_SIZE_KiB_per_launch"] * 1024 * fetch_factor
write_b = out["WRITE_SIZE_KiB_per_launch"] * 1024
out["hbm_read_bytes_per_launch_corrected"] = fetch_b
out["hbm_write_bytes_per_launch"] = write_b
out["traffic_bytes_per_launch"] = fetch_b + write_b
bench = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
out["bench_line_under_profiler"] = bench
out["kernel_source_sha"] = bench.get("kernel_source_sha")  # bench.py quotes this profile's traffic only for the same kernels
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / alg
out["achieved_GBs_from_trace_avg"] = alg / out["avg_ns"]
os.makedirs("profiles", exist_ok=True)
base = os.path.join("profiles", "r%s_%s" % (rnd, w))
import shutil
shutil.copyfile(os.path.join(src, "trace", "t_kernel_stats.csv"), base + "_kernel_stats.csv")  # rocprofv3's own table, verbatim
json.dump(out, open(base + ".json", "w"), indent=1)
with open(base + ".md", "w") as f:
    f.write("# rocprofv3 summary, round %s, workload %s\n\n" % (rnd, w))
    f.write("command: `rocprofv3 --kernel-trace --stats -- python bench.py --workload %s --steps 20 --warmup 3` (+ separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes)\n\n" % w)
    f.write("| kernel | calls | avg us | min us | max us |\n|---|---|---|---|---|\n")
    f.write("| `%s` | %d | %.1f | %.1f | %.1f |\n\n" % (out["kernel"][:90], out["calls"], out["avg_ns"] / 1e3, out["min_ns"] / 1e3, out["max_ns"] / 1e3))
    f.write("* algorithmic bytes per launch: %d -> %.0f GB/s at the trace's average duration (%.1f %% of 8 TB/s)\n" % (alg, out["achieved_GBs_from_trace_avg"], out["achieved_GBs_from_trace_avg"] / 80))
    f.write("* HBM read  (%.3f x FETCH_SIZE x 1024): %.4g B per launch\n* HBM write (WRITE_SIZE x 1024, uncalibrated): %.4g B per launch\n" % (fetch_factor, fetch_b, write_b))
    if "ea_read_bytes_per_launch" in out:
        f.write("* cross-check, L2->fabric read requests x size (TCC_EA0_RDREQ_*): %.4g B per launch\n" % out["ea_read_bytes_per_launch"])
    f.write("* traffic / algorithmic = %.3f\n* workgroup %s, grid %s (registers: the code object's .vgpr_count, scripts/kernel_resources.py -- rocprofv3's VGPR_Count column, %s, is another quantity)\n" % (out["traffic_over_algorithmic"], out["workgroup"], out["grid"], out["rocprof_VGPR_Count_column"]))
print(json.dumps({k: v for k, v in out.items() if k != "bench_line_under_profiler"}, indent=1))
