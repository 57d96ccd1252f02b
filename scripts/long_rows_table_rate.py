"""Few long rows, table-mode automaton: speculative stripes vs one row per lane (NEEDLE_LONG_ROWS=0)."""
import sys, torch
sys.path.insert(0, ".")
from needle_amd.pattern import DFACompiler, unpack_bitmap
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rx = sys.argv[3] if len(sys.argv) > 3 else "sherlock|holmes|watson|irene|adler|john|baker"
p = DFACompiler.compile(rx, "d")
rows = torch.randint(97, 123, (n, mib << 20), dtype=torch.uint8, device="cuda")
rows[:, ::97] = 10
for op, name in ((p.contained_in_batch, "containedIn"), (p.find_batch, "find")):
    for _ in range(2): r = op(rows)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): r = op(rows)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    extra = "" if name != "find" else " first %s %s" % (r[1].tolist()[:2], r[2].tolist()[:2])
    print("%d rows x %d MiB  %-12s %.3f ms  %.1f GB/s%s" % (n, mib, name, ms, rows.numel() / ms / 1e6, extra))
