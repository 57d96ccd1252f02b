import sys, torch, numpy as np
sys.path.insert(0, ".")
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler
n = 10_000_000
rows = torch.empty((n, 256), dtype=torch.uint8, device="cuda")
for s in range(0, n, 1 << 19):
    m = min(1 << 19, n - s); rows[s:s+m] = W.digits_batch(torch, s, m, 256, device="cuda")
lens = (torch.arange(n, device="cuda", dtype=torch.int64) * 2654435761 % 256 + 1).to(torch.int32)
p = DFACompiler.compile("[0-9]+", "d")
for name, op in (("containedIn", p.contained_in_batch), ("matches", p.matches_batch), ("find", p.find_batch)):
    for l, tag in ((None, "full"), (lens, "ragged[1,256]")):
        for _ in range(3): op(rows, l)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): op(rows, l)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%-12s %-14s %.3f ms  %.0f GB/s (row bytes at nominal length)" % (name, tag, ms, n * 256 / ms / 1e6))
