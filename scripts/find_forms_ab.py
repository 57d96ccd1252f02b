"""find() by the lengths automaton vs forward + backward walks (NEEDLE_FIND_LENGTHS=0/1/2): python scripts/find_forms_ab.py"""
import sys, torch
sys.path.insert(0, ".")
import bench
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler
n = 10_000_000
rows = bench.make_rows("c3", W.keywords(1000), 0, n, "cuda")
pats = [("names7", "Sherlock|Holmes|Watson|Irene|Adler|John|Baker"), ("kw12", "|".join(W.keywords(12))), ("kw40", "|".join(W.keywords(40))),
        ("kw300", "|".join(W.keywords(300))), ("kw1000", "|".join(W.keywords(1000)))]
for tag, rx in pats:
    p = DFACompiler.compile(rx, "k")
    pi = p.program_info("forwards", 1)
    out = []
    for r, name in ((rows, "256B"), (rows.view(-1, 32), "32B")):
        for _ in range(3): p.find_batch(r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): p.find_batch(r)
        e1.record(); torch.cuda.synchronize()
        out.append("%s %.3f ms" % (name, e0.elapsed_time(e1) / 10))
    print("%-7s mode %d lengths_form %d states %d lds %d  %s" % (tag, pi["mode"], pi["lengths_form"], pi["n_states"], pi["lds_bytes"], "  ".join(out)))
