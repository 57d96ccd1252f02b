#!/usr/bin/env python3
"""The big-automaton modes on text that keeps the lanes DEEP in the automaton: rows made of the dictionary's own keywords
back to back (separated by one space), 10M x 256 -- the opposite extreme of bench.py's c3s (uniform random letters, where 95 %
of the steps are in states of depth <= 2).  find() then matches in the first chars of every row, so the figure that says what
the table walk costs is containedIn/matches-free: `matches()` never matches here and `find` on a dictionary of words that do
NOT occur walks whole rows: the haystack is built from a SECOND dictionary sharing 4-char prefixes with the first.
Usage: python scripts/dense_dictionary.py [rows] [keywords]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler, unpack_bitmap

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n_kw = int(sys.argv[2]) if len(sys.argv) > 2 else 1000  # 3000: the dictionary at the reference's state limit (bench.py c3x)
words = W.keywords(n_kw, min_len=6, max_len=8)
# decoys: the first 5 chars of a keyword + 2 other chars -- never a keyword, but they drag the automaton to depth 5
rng = np.random.default_rng(11)
kw = set(words)
decoys = []
while len(decoys) < 4096:
    w = words[int(rng.integers(len(words)))]
    d = w[:5] + "".join(chr(97 + int(c)) for c in rng.integers(0, 26, size=2))
    if not any(k in d for k in kw) :
        decoys.append(d)
# every row: decoys back to back separated by spaces (8 chars per slot: 32 slots of 8 = 256)
slot = np.zeros((len(decoys), 8), dtype=np.uint8) + 32
for i, d in enumerate(decoys):
    slot[i, :len(d)] = np.frombuffer(d.encode(), dtype=np.uint8)
slot_t = torch.from_numpy(slot).cuda()
idx = torch.randint(0, len(decoys), (n, 32), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
rows = slot_t[idx].reshape(n, 256).contiguous()
del idx
p = DFACompiler.compile("|".join(words), "t", 0)
mode = p.info()["kernel_mode"]["forwards"]
for name, op in (("find", p.find_batch), ("containedIn", p.contained_in_batch)):
    for _ in range(3):
        r = op(rows)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        r = op(rows)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    words_bm = r[0] if isinstance(r, tuple) else r
    matched = int(unpack_bitmap(words_bm, n).sum())
    print("dense-dictionary text, kernel mode %d: %-11s %.3f ms  %.0f GB/s  matched rows %d" % (mode, name, ms, n * 256 / ms / 1e6, matched))
