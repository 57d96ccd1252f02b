#!/usr/bin/env python3
"""f-4 (round 4): the n-gram candidate filter ALONE (needle_amd/csrc/ngram_probe.hip) over the C3-sparse batch -- time, GB/s
and the windows that pass, against a numpy count on a sample.  Usage: ngram_probe.py [rows] [bits_log2] [stride]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from needle_amd.build import build_probe
from needle_amd import workload

n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 18
S = int(sys.argv[3]) if len(sys.argv) > 3 else 2
M1, M2 = 0xB5297B, 0x68E31D


class NgramParams(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint32) for k in ("on", "stride", "warm", "m1", "m2", "addr_shift", "addr_mask", "bm_bytes", "min_len", "n_grams")]


def h(x):
    x = x.astype(np.uint64)
    return (((x & 0xFFFFFF) * M1 + ((x >> 16) & 0xFFFF) * M2) & 0xFFFFFFFF)


def idx(u):
    return (((u >> (32 - (B - 5))) << 5) | (u & 31)).astype(np.int64)


words = workload.keywords(1000, min_len=6, max_len=8)
grams = set()
for w in words:
    for o in range(S):
        if o + 4 <= len(w):
            grams.add(int.from_bytes(w[o:o + 4].encode(), "little"))
g = np.array(sorted(grams), dtype=np.uint64)
bits = np.zeros(1 << B, dtype=bool)
bits[idx(h(g))] = True
bm = np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype("<u4").reshape(-1)  # bit i of word w = bits[w * 32 + i]
assert all(((bm[i >> 5] >> (i & 31)) & 1) for i in idx(h(g))[:50])
L = ctypes.CDLL(build_probe())
L.ngram_filter_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.POINTER(NgramParams), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
np_ = NgramParams(1, S, 7, M1, M2, 32 - (B - 5) - 2, ((1 << B) // 8 - 1) & ~3, (1 << B) // 8, 6, len(g))
dev = "cuda"
rows = torch.empty((n_rows, 256), dtype=torch.uint8, device=dev)
CH = 1 << 20
for r0 in range(0, n_rows, CH):
    n = min(CH, n_rows - r0)
    rows[r0:r0 + n] = workload.keyword_batch(torch, words, r0, n, 256, device=dev)
d_bm = torch.from_numpy(bm.astype(np.int32)).to(dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
n_units = n_rows * 256 // 1024
blocks = torch.cuda.get_device_properties(0).multi_processor_count


def run():
    rc = L.ngram_filter_probe_launch(rows.data_ptr(), n_units, d_bm.data_ptr(), ctypes.byref(np_), cnt.data_ptr(), blocks, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


run()
torch.cuda.synchronize()
got = int(cnt.item())
# numpy count on the first 64 * 200 rows (whole groups): windows ending inside each row at even ends 2 .. 256 (the first one of a row reaches into the row before)
ns = min(n_rows, 64 * 200)
flat = rows[:ns].cpu().numpy().reshape(-1).astype(np.uint32)
flat = np.concatenate([np.zeros(4, np.uint32), flat])
ends = np.arange(S, ns * 256 + 1, S) + 4
four = flat[ends - 4] | (flat[ends - 3] << 8) | (flat[ends - 2] << 16) | (flat[ends - 1] << 24)
want_s = int(bits[idx(h(four))].sum())
cnt.zero_()
rc = L.ngram_filter_probe_launch(rows.data_ptr(), ns * 256 // 1024, d_bm.data_ptr(), ctypes.byref(np_), cnt.data_ptr(), blocks, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
got_s = int(cnt.item())
print("sample: device %d numpy %d (may differ by the windows that straddle a 16 KiB group start: <= %d)" % (got_s, want_s, ns // 64))
ts = []
for _ in range(10):
    cnt.zero_()
    torch.cuda.synchronize()
    t = time.perf_counter()
    run()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("rows %d B %d S %d grams %d fill %.5f: %.4f ms (wall best %.4f) = %.0f GB/s, windows passed %d = %.3f per row" % (
    n_rows, B, S, len(g), bits.mean(), ms, min(ts) * 1e3, n_rows * 256 / ms / 1e6, got, got / n_rows))
