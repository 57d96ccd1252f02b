"""Ragged-row cost of the big-table kernels: the C3 dictionary over the C3 batch with no lengths, lengths = 256 everywhere
(the guarded kernels on full rows) and lengths uniform in [1, 256]."""
import sys, torch
sys.path.insert(0, ".")
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"  # c3 | c3s
p, _, words = bench.make_pattern(wl)
rows = bench.make_rows(wl, words, 0, n, "cuda:0")
lens = (torch.arange(n, device="cuda", dtype=torch.int64) * 2654435761 % 256 + 1).to(torch.int32)
full_lens = torch.full((n,), 256, dtype=torch.int32, device="cuda")
for name, op in (("containedIn", p.contained_in_batch), ("find", p.find_batch)):
    for l, tag in ((None, "full"), (full_lens, "lengths=256"), (lens, "ragged[1,256]")):
        for _ in range(3): op(rows, l)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): op(rows, l)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%s %-12s %-14s %.3f ms" % (wl, name, tag, ms))
