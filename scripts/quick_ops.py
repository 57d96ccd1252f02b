"""Times all three ops of one bench workload on the resident batch: python scripts/quick_ops.py c3"""
import sys, torch
sys.path.insert(0, ".")
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "c3"
p, what, words = bench.make_pattern(w)
n = 10_000_000
rows = bench.make_rows(w, words, 0, n, "cuda")
cw = rows.element_size()
for op, name, extra in ((p.contained_in_batch, "containedIn", 0), (p.matches_batch, "matches", 0), (p.find_batch, "find", 8)):
    for _ in range(3): op(rows)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): r = op(rows)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(w, name, "ms %.4f" % ms, "GB/s %.0f" % (n * (256 * cw + extra) / ms / 1e6))
