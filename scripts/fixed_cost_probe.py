"""Fixed cost of a filter-kernel launch (LDS staging of the automaton + bitmaps by every CU, launch, tail): find() of the c3s dictionary on
batches of growing size, HIP-event time per launch -> intercept and slope.  python scripts/fixed_cost_probe.py [workload]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "c3s"
p, label, words = bench.make_pattern(w)
dev = "cuda"
rows_all = bench.make_rows(w, words, 0, 2_500_000, dev)
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (4096, 65536, 262144, 1_000_000, 2_500_000)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for n in sizes:
    rows = rows_all[:n]
    words_o = torch.empty((n + 63) // 64, dtype=torch.int64, device=dev)
    se = torch.empty(n, dtype=torch.int32, device=dev)
    for _ in range(3):
        p.find_packed16_batch(rows, out=(words_o, se))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        p.find_packed16_batch(rows, out=(words_o, se))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{w} rows {n:8d}: {ms*1000:8.1f} us per call, {n*256/ms*1e-6:7.1f} GB/s", flush=True)
