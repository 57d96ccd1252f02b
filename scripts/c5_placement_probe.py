#!/usr/bin/env python3
"""C5 `find` runs at 0.87 ms per launch in some processes and 0.95 ms in others on the same box in the same lease (fresh processes, same
binary, same batch: bench.py runs of round 5).  Inside ONE process the time is stable.  What differs between processes is where the driver
put the 5.12 GB batch (and the 80 MB of results).  This probe holds that fixed point by point: one process, the SAME rows copied into several
separately allocated buffers (all kept alive), the kernel timed on each; then one rows buffer with several separately allocated result sets.
Usage: python scripts/c5_placement_probe.py [workload] [buffers]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

w = sys.argv[1] if len(sys.argv) > 1 else "c5"
nbuf = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pattern, label, words = bench.make_pattern(w)
rows0 = bench.make_rows(w, words, 0, 10_000_000, "cuda:0")
n = rows0.shape[0]
op = pattern.contained_in_batch if w == "c2" else pattern.find_batch
is_find = w != "c2"


def outs():
    bm = torch.empty((n + 63) // 64, dtype=torch.int64, device="cuda:0")
    return (bm, torch.empty(n, dtype=torch.int32, device="cuda:0"), torch.empty(n, dtype=torch.int32, device="cuda:0")) if is_find else bm


def timed(rows, out, k=40):
    for _ in range(5):
        op(rows, out=out)
    torch.cuda.synchronize()
    evs = []
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); op(rows, out=out); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2], ms[0], ms[-1]


print(w, label)
out0 = outs()
keep = [rows0]
for i in range(nbuf):
    r = rows0 if i == 0 else torch.empty_like(rows0)
    if i:
        r.copy_(rows0)
        keep.append(r)
    med, lo, hi = timed(r, out0)
    # (does a plain streaming read of the buffer see the same thing?  bench.py's read-ceiling probe on this buffer)
    print("rows buffer %d at 0x%x (results fixed): median %.4f ms  min %.4f  max %.4f   streaming read of it: %.0f GB/s" %
          (i, r.data_ptr(), med, lo, hi, bench.measured_read_ceiling(r) or 0.0))
keep_o = [out0]
for i in range(nbuf):
    o = outs()
    keep_o.append(o)
    med, lo, hi = timed(rows0, o)
    print("result set %d at 0x%x (rows buffer 0): median %.4f ms  min %.4f  max %.4f" % (i + 1, (o[0] if is_find else o).data_ptr(), med, lo, hi))
med, lo, hi = timed(rows0, out0)
print("rows buffer 0, result set 0 again: median %.4f ms" % med)
