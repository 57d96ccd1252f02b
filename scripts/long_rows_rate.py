"""Throughput of the stripe path on few, long rows: python scripts/long_rows_rate.py [n_rows] [MiB per row]"""
import sys, torch
sys.path.insert(0, ".")
from needle_amd.pattern import DFACompiler, unpack_bitmap
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 256
p = DFACompiler.compile("[0-9]+", "d")
rows = torch.randint(97, 123, (n, mib << 20), dtype=torch.uint8, device="cuda")
rows[:, -7:-3] = 53  # one digit run at the very end of every row
for op, name in ((p.contained_in_batch, "containedIn"), (p.matches_batch, "matches"), (p.find_batch, "find")):
    for _ in range(2): r = op(rows)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): r = op(rows)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    extra = "" if name != "find" else " start/end %s %s" % (r[1].tolist()[:2], r[2].tolist()[:2])
    print("%d rows x %d MiB  %-12s %.3f ms  %.0f GB/s%s" % (n, mib, name, ms, rows.numel() / ms / 1e6, extra))
