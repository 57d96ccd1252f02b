import sys, time, torch, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import load_snapshot
from test_gpu_parity import pattern_from_fixture
from needle_amd import workload as W
p = pattern_from_fixture(load_snapshot("DigitPlus"))
n = 10_000_000
rows = torch.empty((n,256), dtype=torch.uint8, device="cuda")
for s in range(0, n, 1<<19):
    m = min(1<<19, n-s); rows[s:s+m] = W.digits_batch(torch, s, m, 256, device="cuda")
for op,name,extra in ((p.contained_in_batch,"containedIn",0),(p.matches_batch,"matches",0),(p.find_batch,"find",8)):
    for _ in range(3): op(rows)
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): r = op(rows)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/10
    print(name, "ms", ms, "GB/s", n*(256+extra)/ms/1e6)
