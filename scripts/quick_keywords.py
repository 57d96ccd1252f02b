"""Throughput vs dictionary size (LDS table -> HBM/L2 table cliff): python scripts/quick_keywords.py <n_keywords>"""
import sys, torch
sys.path.insert(0, ".")
import bench
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler, unpack_bitmap
nk = int(sys.argv[1])
words = W.keywords(nk)
p = DFACompiler.compile("|".join(words), "k")
n = 10_000_000
rows = bench.make_rows("c3", W.keywords(1000), 0, n, "cuda")
print(nk, "keywords", p.info()["n_states"]["forwards"], "states, mode", p.info()["kernel_mode"]["forwards"])
for op, name in ((p.contained_in_batch, "containedIn"), (p.find_batch, "find")):
    for _ in range(2): r = op(rows)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): r = op(rows)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("  %-12s %.3f ms  %.0f GB/s" % (name, ms, n * 256 / ms / 1e6))
