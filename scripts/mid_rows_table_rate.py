"""Medium-long rows, table-mode automaton: python scripts/mid_rows_table_rate.py <n_rows> <KiB per row>"""
import sys, torch
sys.path.insert(0, ".")
from needle_amd.pattern import DFACompiler
n, kib = int(sys.argv[1]), int(sys.argv[2])
p = DFACompiler.compile("sherlock|holmes|watson|irene|adler|john|baker", "d")
rows = torch.randint(97, 123, (n, kib << 10), dtype=torch.uint8, device="cuda")
rows[:, ::97] = 10
rows[::2, :] = torch.where(rows[::2, :] == 115, torch.full_like(rows[::2, :], 120), rows[::2, :])  # no 's' in even rows: fewer early hits
for op, name in ((p.contained_in_batch, "containedIn"), (p.find_batch, "find"), (p.matches_batch, "matches")):
    for _ in range(2): r = op(rows)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): r = op(rows)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("%d rows x %d KiB  %-12s %.3f ms  %.1f GB/s" % (n, kib, name, ms, rows.numel() / ms / 1e6))
