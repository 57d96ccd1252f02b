import sys, torch, numpy as np
sys.path.insert(0, ".")
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler
n = 10_000_000
rows = torch.empty((n, 256), dtype=torch.uint8, device="cuda")
for s in range(0, n, 1 << 19):
    m = min(1 << 19, n - s); rows[s:s+m] = W.digits_batch(torch, s, m, 256, device="cuda")
lens = (torch.arange(n, device="cuda", dtype=torch.int64) * 2654435761 % 256 + 1).to(torch.int32)
rx = sys.argv[1] if len(sys.argv) > 1 else "[0-9]+"
p = DFACompiler.compile(rx, "d")
full_lens = torch.full((n,), 256, dtype=torch.int32, device="cuda")
print(rx)
for name, op in (("containedIn", p.contained_in_batch), ("matches", p.matches_batch), ("find", p.find_batch)):
    for l, tag in ((None, "full"), (full_lens, "lengths=256"), (lens, "ragged[1,256]")):
        for _ in range(3): op(rows, l)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): op(rows, l)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%-12s %-14s %.3f ms  %.0f GB/s (row bytes at nominal length)" % (name, tag, ms, n * 256 / ms / 1e6))
