"""Throughput on many SHORT rows: python scripts/short_rows_rate.py <stride> [regex]"""
import sys, torch
sys.path.insert(0, ".")
from needle_amd.pattern import DFACompiler, unpack_bitmap
stride = int(sys.argv[1]); rx = sys.argv[2] if len(sys.argv) > 2 else "[0-9]+"
n = 2_560_000_000 // stride
p = DFACompiler.compile(rx, "d")
rows = torch.randint(97, 123, (n, stride), dtype=torch.uint8, device="cuda")
rows[::3, stride // 2] = 53
for op, name in ((p.contained_in_batch, "containedIn"), (p.find_batch, "find"), (p.find_packed16_batch, "find (one dword per row)"),
                 (p.find_packed8_batch, "find (one uint16 per row)")):  # round 6: needle_find_packed8_dev
    for _ in range(2): r = op(rows)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): r = op(rows)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    hits = unpack_bitmap(r[0] if isinstance(r, tuple) else r, n).mean()
    print("%d rows x %d B  %-26s %.3f ms  %.0f GB/s  hit %.3f" % (n, stride, name, ms, rows.numel() / ms / 1e6, hits))
