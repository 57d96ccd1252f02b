import sys, torch
sys.path.insert(0, ".")
import bench
p, _, words = bench.make_pattern("c2")
n = 10_000_000
rows = bench.make_rows("c2", words, 0, n, "cuda")
cur = torch.zeros(n, dtype=torch.int32, device="cuda")
cur2 = (torch.arange(n, device="cuda") % 200).to(torch.int32)
for name, c in (("cursor=0", cur), ("cursor=r%200", cur2)):
    for _ in range(3): p.find_next_batch(rows, c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): r = p.find_next_batch(rows, c)
    e1.record(); torch.cuda.synchronize()
    print("find_next %-14s %.3f ms" % (name, e0.elapsed_time(e1) / 10))
import time
t0 = time.perf_counter(); off, s, e = p.find_all_batch(rows); torch.cuda.synchronize()
print("find_all: %.1f ms, %d matches" % ((time.perf_counter() - t0) * 1e3, s.numel()))
for slots in (1, 2, 4):
    p.find_all_dense(rows, slots); torch.cuda.synchronize()
    t0 = time.perf_counter(); counts, ds, de, more = p.find_all_dense(rows, slots); torch.cuda.synchronize()
    print("find_all_dense slots=%d: %.2f ms, %d matches filed, more=%s" % (slots, (time.perf_counter() - t0) * 1e3, int(counts.sum()), more))
