"""PCIe-inclusive rate of the *_host entry points (upload -> kernel -> download), for DESIGN.md; never the bench value."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler
p = DFACompiler.compile("[0-9]+", "d")
n = 2_000_000
rows = W.digits_batch(np, 0, n, 256)
p.contained_in_batch(rows[:1000])
for _ in range(3):
    t0 = time.perf_counter(); p.contained_in_batch(rows); dt = time.perf_counter() - t0
    print("containedIn host path: %d rows, %.1f ms, %.1f GB/s incl. hipMalloc + H2D (pageable) + kernel + D2H" % (n, dt * 1e3, rows.nbytes / dt / 1e9))
