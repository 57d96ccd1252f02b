#!/usr/bin/env python3
"""Round 4, f-4: containedIn() / find() on the C3-sparse batch, kernel time by HIP events; prints popcount and a checksum of
start / end so that runs with NEEDLE_PREFILTER=0 / 1 can be compared.  Usage: r4_ngram.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
words = W.keywords(1000, min_len=6, max_len=8)
p = DFACompiler.compile("|".join(words), "t", 0)
print("prefilter", {k: v for k, v in p.prefilter_info("forwards").items() if k in ("on", "stride", "warm", "n_windows", "bitmap_bytes", "why")})
rows = torch.empty((n, 256), dtype=torch.uint8, device="cuda")
CH = 1 << 20
for r0 in range(0, n, CH):
    m = min(CH, n - r0)
    rows[r0:r0 + m] = W.keyword_batch(torch, words, r0, m, 256, device="cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {}
def f_find():
    out["f"] = p.find_batch(rows)
def f_cont():
    out["c"] = p.contained_in_batch(rows)
tf, tc = timed(f_find), timed(f_cont)
fw, fs, fe = out["f"]
idx = torch.arange(n, device="cuda", dtype=torch.int64) % 65521 + 1
chk = int(((fs.to(torch.int64) + 3 * fe.to(torch.int64) + 7) * idx).sum().item())
pc = lambda w: int(sum(bin(int(x) & 0xFFFFFFFFFFFFFFFF).count("1") for x in w.cpu().numpy().view(np.uint64)))
print("NEEDLE_PREFILTER=%s rows %d: find %.4f ms (%.0f GB/s, %.3f of 8 TB/s)  containedIn %.4f ms (%.0f GB/s)  matched %d / %d  checksum %d" % (
    os.environ.get("NEEDLE_PREFILTER", "1"), n, tf, (n * 264 + n / 8) / tf / 1e6, (n * 264 + n / 8) / tf / 1e6 / 8000, tc, n * 256 / tc / 1e6, pc(fw), pc(out["c"]), chk))
