"""Pair-table automata: workgroup shape A/B (NEEDLE_SHAPE=16x64 ...): python scripts/pair_shape_ab.py"""
import sys, torch
sys.path.insert(0, ".")
import bench
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler
n = 10_000_000
rows = bench.make_rows("c3", W.keywords(1000), 0, n, "cuda")
pats = [("names7", "Sherlock|Holmes|Watson|Irene|Adler|John|Baker"), ("kw12", "|".join(W.keywords(12))), ("a.c", "a.c"), ("ing", "[a-z]+ing"), ("kw40", "|".join(W.keywords(40)))]
for tag, rx in pats:
    p = DFACompiler.compile(rx, "k")
    pi = p.program_info("forwards", 1)
    out = []
    for name, op in (("find", p.find_batch), ("containedIn", p.contained_in_batch), ("matches", p.matches_batch)):
        for _ in range(3): op(rows)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): op(rows)
        e1.record(); torch.cuda.synchronize()
        out.append("%s %.3f" % (name, e0.elapsed_time(e1) / 10))
    print("%-7s mode %d lds %d shape %dx%d  %s" % (tag, pi["mode"], pi["lds_bytes"], pi["waves"], pi["tile_bytes"], "  ".join(out)))
