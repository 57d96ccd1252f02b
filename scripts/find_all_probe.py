#!/usr/bin/env python3
"""Time needle_find_all_dev (FIND_ALL_PROBE_PACKED=1: needle_find_all_packed16_dev) on a bench workload's rows (every non-overlapping match of every row):
python scripts/find_all_probe.py <c2|c3|c3s|c5> [rows] [slots] [check]
NEEDLE_FIND_ALL_ROUNDS=1: the round-per-match form; NEEDLE_FIND_ALL_DEFER=0: backward walks at once.
check: compare a sample of rows with the oracle's repeated find()."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

w = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 32
pattern, label, words = bench.make_pattern(w)
rows = bench.make_rows(w, words, 0, n, "cuda:0")
cw = rows.element_size()
torch.cuda.synchronize()
out = {"workload": w, "rows": n, "slots": slots, "rounds": os.environ.get("NEEDLE_FIND_ALL_ROUNDS", "0"),
       "defer": os.environ.get("NEEDLE_FIND_ALL_DEFER", "1")}
best = None
counts = torch.zeros(n, dtype=torch.int32, device="cuda:0")
st = torch.full((n, slots), -1, dtype=torch.int32, device="cuda:0")
en = torch.full((n, slots), -1, dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()
# device pre-warm as in bench.py: an idle GPU runs its first ~40 ms of load below its steady clocks
scratch = torch.empty_like(rows)
tp = time.perf_counter()
while time.perf_counter() - tp < 0.15:
    for _ in range(8):
        scratch.copy_(rows)
    torch.cuda.synchronize()
del scratch
packed = bool(os.environ.get("FIND_ALL_PROBE_PACKED"))  # needle_find_all_packed16_dev: one dword per match
blocked = bool(os.environ.get("FIND_ALL_PROBE_BLOCKED"))  # needle_find_all_blocked16_dev: one dword per match, group-blocked slots
if blocked:
    st = torch.full(((n + 63) // 64, slots, 64), -1, dtype=torch.int32, device="cuda:0")
out["form"] = "blocked16" if blocked else "packed16" if packed else "start/end int32"
for rep in range(4):
    t0 = time.perf_counter()
    if blocked:
        counts, st, more = pattern.find_all_blocked16(rows, slots, out=(counts, st))
    elif packed:
        counts, st, more = pattern.find_all_dense_packed16(rows, slots, out=(counts, st))
    else:
        counts, st, en, more = pattern.find_all_dense(rows, slots, out=(counts, st, en))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    best = dt if best is None or (rep and dt < best) else best
if blocked:
    st = pattern.unblock16(st, n)
if packed or blocked:
    st, en = st & 0xFFFF, (st >> 16) & 0xFFFF
out["ms"] = round(best * 1e3, 3)
total = int(counts.sum().item())
out["matches"] = total
out["max_per_row"] = int(counts.max().item())
out["more"] = bool(more)
out["GB/s"] = round(n * 256 * cw / best / 1e9, 1)
out["matches/s"] = round(total / best / 1e6, 1)
# compact form: count pass + fill pass (find_all_csr also allocates and prefix-sums: timed apart)
for rep in range(0 if os.environ.get("FIND_ALL_PROBE_DENSE_ONLY") else 3):
    t0 = time.perf_counter()
    cnt = pattern.count_matches_batch(rows)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["count_ms"] = round(dt * 1e3, 3) if rep == 0 or dt * 1e3 < out["count_ms"] else out["count_ms"]
for rep in range(0 if os.environ.get("FIND_ALL_PROBE_DENSE_ONLY") else 3):
    t0 = time.perf_counter()
    offs, cs, ce = pattern.find_all_csr(rows)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["csr_ms"] = round(dt * 1e3, 3) if rep == 0 or dt * 1e3 < out["csr_ms"] else out["csr_ms"]
for rep in range(0 if os.environ.get("FIND_ALL_PROBE_DENSE_ONLY") else 3):  # the compact form in one call (blocked pass + scan + compaction)
    t0 = time.perf_counter()
    o2, se2, _ = pattern.find_all_compact16(rows, slots, cap=total + 16, want_more=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["compact16_ms"] = round(dt * 1e3, 3) if rep == 0 or dt * 1e3 < out["compact16_ms"] else out["compact16_ms"]
if not os.environ.get("FIND_ALL_PROBE_DENSE_ONLY"):
    assert int(o2[-1].item()) == total and len(se2) == total
    assert int(offs[-1].item()) == total or more
if len(sys.argv) > 4 and sys.argv[4] == "check":
    import numpy as np
    from oracle.walker import Dfa, OraclePattern
    t = pattern.tables()
    d = {k: Dfa(t["class_map"], t["stride"], v["table"], v["accepting"], v["max_char"]) for k, v in t["dfas"].items()}
    o = OraclePattern(d["matches"], d["contained_in"], d["forwards"], d["backwards"], t["fixed_len"], -1)
    idx = np.linspace(0, n - 1, 300).astype(np.int64)
    hr = rows[idx].cpu().numpy()
    hc, hs, he = counts[idx].cpu().numpy(), st[idx].cpu().numpy(), en[idx].cpu().numpy()
    bad = 0
    for i in range(len(idx)):
        want = o.find_all(hr[i].view(np.uint16) if cw == 2 else hr[i])
        got = [(int(hs[i, k]), int(he[i, k])) for k in range(int(hc[i]))]
        if got != [tuple(x) for x in want][:slots]:
            bad += 1
            if bad < 3:
                print("MISMATCH row", idx[i], got[:6], want[:6])
    out["checked"] = len(idx)
    out["bad"] = bad
print(json.dumps(out))
