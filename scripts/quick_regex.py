"""Times the three ops of an arbitrary regex on a bench batch: python scripts/quick_regex.py <c2|c3|c5> <regex>"""
import sys, torch
sys.path.insert(0, ".")
import bench
from needle_amd.pattern import DFACompiler, unpack_bitmap
w, rx = sys.argv[1], sys.argv[2]
_, _, words = bench.make_pattern(w)
p = DFACompiler.compile(rx, "q")
n = 10_000_000
rows = bench.make_rows(w, words, 0, n, "cuda")
cw = rows.element_size()
print(rx, p.info()["n_states"], "mode", p.info()["kernel_mode"])
for op, name, extra in ((p.contained_in_batch, "containedIn", 0), (p.matches_batch, "matches", 0), (p.find_batch, "find", 8)):
    for _ in range(3): r = op(rows)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): r = op(rows)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    hits = unpack_bitmap(r[0] if isinstance(r, tuple) else r, n).mean()
    print("  %-12s ms %.4f  GB/s %.0f  hit rate %.3f" % (name, ms, n * (256 * cw + extra) / ms / 1e6, hits))
