#!/usr/bin/env python3
"""Measures the empirical HBM streaming-READ ceiling on this GPU with a trivial read-reduce kernel (BASELINE.md s2:
report it beside the nominal 8 TB/s).  Builds scripts/stream_read.hip with hipcc into /tmp."""
import ctypes, json, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/stream_read.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "..", "needle_amd", "csrc", "stream_probe.hip")])
L = ctypes.CDLL(so)
L.stream_read_launch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
n = 2_560_000_000
buf = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
out = torch.zeros(4, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
best = {}
for blocks in (256 * 4, 256 * 8, 256 * 16, 256 * 32):
    for unroll in (1, 4, 8, 104, 108):
        for _ in range(3):
            L.stream_read_launch(buf.data_ptr(), n, out.data_ptr(), blocks, unroll, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.stream_read_launch(buf.data_ptr(), n, out.data_ptr(), blocks, unroll, s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        best[(blocks, unroll)] = n / ms / 1e6
        print(blocks, unroll, "%.3f ms  %.0f GB/s" % (ms, n / ms / 1e6))
k = max(best, key=best.get)
print(json.dumps({"stream_read_ceiling_GBs": best[k], "blocks": k[0], "unroll": k[1], "bytes": n}))
