#!/usr/bin/env python3
"""gpurun_out/prof_fa (scripts/profile_find_all.sh) -> profiles/r03_find_all.md + the per-workload kernel-stats tables."""
import csv
import json
import os
import re
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "04"
src = os.path.join(root, "gpurun_out", "prof_fa")
out = []
out.append("# needle::find_all_kernel, round %d (`scripts/profile_find_all.sh`: `rocprofv3 --kernel-trace --stats -- python scripts/find_all_probe.py <workload> 10000000 32`)\n" % int(RND))
out.append("Every non-overlapping match of every row (the reference's repeated `Matcher.find()`), 10⁷ × 256-char rows resident in HBM, 32 result slots per row, "
           "outputs preallocated.  `probe ms` = host-timed call + synchronise (best of 4); `kernel µs` = rocprofv3 average of the kernel; "
           "`rounds ms` = the round-per-match form on the same rows (`NEEDLE_FIND_ALL_ROUNDS=1`: one scan of the batch and one stream synchronisation per round).\n")
out.append("| workload | matches | busiest row | probe ms | kernel µs (calls) | G matches/s | input GB/s | rounds ms | speed-up | count pass ms | compact form ms (count + prefix sum + fill + allocations) |")
out.append("|---|---|---|---|---|---|---|---|---|---|---|")
for w in ("c3", "c2", "c5", "c3s"):
    one = json.loads(open(os.path.join(src, w + ".json")).read().strip().splitlines()[-1])
    rnd = json.loads(open(os.path.join(src, w + "_rounds.json")).read().strip().splitlines()[-1])
    k_avg, calls = 0.0, None
    stats = os.path.join(src, w, "t_kernel_stats.csv")
    for r in csv.DictReader(open(stats)):
        if "find_all_kernel" in r["Name"] or "ngram_kernel" in r["Name"]:  # (c3s since round 4: the filter kernel's find-all form)
            k_avg += float(r["AverageNs"]) / 1e3
            calls = int(r["Calls"]) if calls is None else calls
    shutil.copy(stats, os.path.join(root, "profiles", "r%s_find_all_%s_kernel_stats.csv" % (RND, w)))
    out.append("| %s | %d | %d | %.3f | %.1f (%d) | %.1f | %.0f | %.2f | %.1fx | %.2f | %.2f |" % (
        w, one["matches"], one["max_per_row"], one["ms"], k_avg, calls, one["matches"] / (k_avg * 1e-6) / 1e9,
        one["rows"] * 256 * (2 if w == "c5" else 1) / (k_avg * 1e-6) / 1e9, rnd["ms"], rnd["ms"] / one["ms"], one.get("count_ms", 0), one.get("csr_ms", 0)))
# the dictionary with one dword per match (needle_find_all_packed16_dev)
pk = json.loads(open(os.path.join(src, "c3_packed.json")).read().strip().splitlines()[-1])
pk_k = sum(float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(os.path.join(src, "c3_packed", "t_kernel_stats.csv"))) if "find_all_kernel" in r["Name"])
shutil.copy(os.path.join(src, "c3_packed", "t_kernel_stats.csv"), os.path.join(root, "profiles", "r%s_find_all_c3_packed16_kernel_stats.csv" % RND))
pk_t = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    pk_t[c] = float(open(os.path.join(src, "packed_%s.txt" % c)).read().split()[1])
pk_bytes = pk_t["FETCH_SIZE"] * 2048 + pk_t["WRITE_SIZE"] * 1024
out.append("\n`needle_find_all_packed16_dev` (each match one dword, start | end << 16; the same kernel, one store per match) on c3: probe %.3f ms, kernel %.1f µs, "
           "%d matches (%d of 300 sampled rows differ from the oracle's repeated find()); FETCH_SIZE %.2f GB + WRITE_SIZE %.2f GB = **%.2f GB per launch** "
           "= %.2f x (rows + 4 B per row + 4 B per match).\n" % (pk["ms"], pk_k, pk["matches"], pk.get("bad", -1), pk_t["FETCH_SIZE"] * 2048 / 1e9, pk_t["WRITE_SIZE"] * 1024 / 1e9,
                                                               pk_bytes / 1e9, pk_bytes / (1e7 * 260 + 4 * pk["matches"])))
pmc = {}
for line in open(os.path.join(src, "pmc_c3.txt")):
    m = re.match(r"(\w+)\s+([0-9.e+]+)$", line.strip())
    if m:
        pmc[m.group(1)] = float(m.group(2))
    m = re.match(r"kernel dur us \(under pmc\): ([0-9.]+)", line)
    if m:
        pmc["kernel_us"] = float(m.group(1))
traffic = {}
for d, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rows = [r for r in csv.DictReader(open(os.path.join(src, d, "p_counter_collection.csv"))) if r["Counter_Name"] == name]
    calls = max(1, sum(1 for r in rows if "find_all_kernel" in r["Kernel_Name"]))
    traffic[name] = sum(float(r["Counter_Value"]) for r in rows) / calls
cu_cycles = pmc["GRBM_GUI_ACTIVE"] / 8 * 256
cw = 4e7
out.append("\n## PMC, C3 dictionary (one counter group per pass, `scripts/pmc_find_all.sh`)\n")
out.append("| counter | per launch |")
out.append("|---|---|")
for k in ("kernel_us", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
          "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
    if k in pmc:
        out.append("| `%s` | %.4g |" % (k, pmc[k]))
out.append("| `FETCH_SIZE` (KiB; x 2 on gfx950, MI355X_MICROARCH.md) | %.4g -> %.2f GB read |" % (traffic["FETCH_SIZE"], traffic["FETCH_SIZE"] * 1024 * 2 / 1e9))
out.append("| `WRITE_SIZE` (KiB) | %.4g -> %.2f GB written |" % (traffic["WRITE_SIZE"], traffic["WRITE_SIZE"] * 1024 / 1e9))
out.append("\nDerived (CU-cycles = GRBM_GUI_ACTIVE / 8 x 256 CUs; 4e7 char-waves per launch):\n")
out.append("| | |")
out.append("|---|---|")
out.append("| VALU busy (4 cycles per wave64 op, 4 SIMDs per CU) | %.0f %% |" % (100 * pmc["SQ_INSTS_VALU"] / cu_cycles))
out.append("| LDS array busy | %.0f %% |" % (100 * pmc["SQ_LDS_IDX_ACTIVE"] / cu_cycles))
out.append("| bank-conflict share of LDS cycles | %.0f %% |" % (100 * pmc["SQ_LDS_BANK_CONFLICT"] / pmc["SQ_LDS_IDX_ACTIVE"]))
out.append("| VALU instructions per char-wave | %.1f (scan kernel, C3 find: 3.8) |" % (pmc["SQ_INSTS_VALU"] / cw))
out.append("| LDS instructions per char-wave | %.2f |" % (pmc["SQ_INSTS_LDS"] / cw))
out.append("| HBM bytes per launch / (rows + 4 B per row + 8 B per match) | %.2f |" % (
    (traffic["FETCH_SIZE"] * 2048 + traffic["WRITE_SIZE"] * 1024) / (1e7 * 260 + 8 * 46586444)))
out.append("\nSince round 3: for the dictionary (a keyword union) the program is the refined \"lengths\" automaton (DESIGN.md s3): the state a search ends in says how long "
           "its match was, start = end - pend[state] -- no starts phase, no backward walks, and FETCH_SIZE is the batch exactly once (round 2: 6.2 GB).  The walk kernel is "
           "VALU-issue bound, not HBM bound: an iteration walks a whole 16-byte piece for every lane under the cursor guard and a tile takes as many iterations as its "
           "busiest lane.  The written bytes are several times the results, and exactly the slot arrays: with 32 slots a row's block is one 128-byte line in each of the two "
           "arrays, almost every row has a match, and a line leaves the L2 whole -- 2 x 128 B x 10^7 rows = 2.56 GB whatever the order of the stores (each line is "
           "written once: staging the stores in LDS would not change the count).  The packed form above halves it -- one array, one line per row.  c2 / c5 / c3s: patterns with unbounded match lengths, or a refined "
           "automaton that does not fit the LDS as a plain table (c3s), keep indexBackwards at the end of every 64-row group.  Round 4: the kernel is unchanged; the compressed lengths program "
           "was built into it for c3s (`NEEDLE_FIND_ALL_LENGTHS=2`, parity-tested) and measured no faster than hot rows + backward walks (2.05 against 1.98 ms: "
           "`profiles/r04_find_all_c3s_ab.log`) -- what a find-all costs there is the per-lane piece walk, not the 0.25 starts per row.  "
           "What c3s takes now is NOT this kernel: dictionaries whose find() runs behind the n-gram candidate filter have their find-all there too "
           "(`needle::ngram_kernel<3, ...>`, needle_ngram.hip: every verified candidate filed with its row, the rows sort theirs out against the moving cursor) -- the c3s "
           "line above is that kernel (`NEEDLE_FIND_ALL_FILTER=0`: 1.99-2.04 ms, count pass 1.75, compact form 3.8).  Host batches cross PCIe in the one-dword form "
           "(`needle_find_all_host`).")
open(os.path.join(root, "profiles", "r%s_find_all.md" % RND), "w").write("\n".join(out) + "\n")
print("\n".join(out))
