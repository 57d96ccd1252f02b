#!/usr/bin/env python3
"""Round 4, f-4: the n-gram filter kernel on text built to defeat it -- rows made of the dictionary's own keywords with the FIRST
char replaced (the windows the filter keys on are the LAST chars of a keyword: every slot is a candidate, almost none is a match) --
against the ordinary scan kernel on the same rows (NEEDLE_PREFILTER=0), and on the bench's own text for reference.
Usage: NEEDLE_PREFILTER=0|1 python scripts/ngram_worstcase.py [rows] [keywords]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler, unpack_bitmap

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n_kw = int(sys.argv[2]) if len(sys.argv) > 2 else 1000  # 3000: the dictionary at the reference's state limit (bench.py c3x)
words = W.keywords(n_kw, min_len=6, max_len=8)
p = DFACompiler.compile("|".join(words), "t", 0)
slot = np.zeros((len(words), 8), dtype=np.uint8) + 32
for i, w in enumerate(words):
    b = np.frombuffer(w.encode(), dtype=np.uint8).copy()
    b[0] = ord("q") if b[0] != ord("q") else ord("z")
    slot[i, 8 - len(b):] = b  # right-aligned in an 8-byte slot: the keyword's tail ends at the slot's end
slot_t = torch.from_numpy(slot).cuda()
idx = torch.randint(0, len(words), (n, 32), device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
rows = slot_t[idx].reshape(n, 256).contiguous()
del idx


def timed(fn, reps=10):
    for _ in range(3):
        r = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        r = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, r


# the very first call on this text (the flood watch has seen nothing yet: the filter kernel runs), then the steady state
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); p.find_batch(rows); e1.record(); torch.cuda.synchronize()
print("first find() on this text: %.3f ms" % e0.elapsed_time(e1))
tf, rf = timed(lambda: p.find_batch(rows))
tc, rc = timed(lambda: p.contained_in_batch(rows))
matched = int(unpack_bitmap(rf[0], n).sum())
chk = int(((rf[1].long() + 3 * rf[2].long() + 7) * (torch.arange(n, device="cuda") % 65521 + 1)).sum().item())
print("NEEDLE_PREFILTER=%s NEEDLE_PREFILTER_WATCH=%s decapitated-keyword text, %d rows: find %.3f ms  containedIn %.3f ms  matched rows %d  checksum %d" % (
    os.environ.get("NEEDLE_PREFILTER", "1"), os.environ.get("NEEDLE_PREFILTER_WATCH", "1"), n, tf, tc, matched, chk))
# a sample against the oracle
from test_compile_matches_txt import oracle_for
o, _ = oracle_for("|".join(words), 0)
k = 20000
host = rows[:k].cpu().numpy()
of, ofs, ofe = o.batch_find(host, threads=8)
assert (unpack_bitmap(rf[0], n)[:k] == of).all() and (rf[1][:k].cpu().numpy() == ofs).all() and (rf[2][:k].cpu().numpy() == ofe).all()
print("oracle sample ok (%d rows, %d matched)" % (k, int(of.sum())))
