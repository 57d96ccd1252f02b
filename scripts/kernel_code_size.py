#!/usr/bin/env python3
"""Code size (bytes of ISA) of every kernel of the library, cross-compiled here (no GPU needed): a CU pair shares a 64 KB
instruction cache, and an instantiation that outgrows it fetches its inner loop from L2.  python scripts/kernel_code_size.py [min_bytes]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "needle_amd", "csrc")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
names = {"0": "matches", "1": "containedIn", "2": "find"}
modes = {"0": "pack", "1": "table8", "2": "table16", "3": "hbm", "4": "pair", "5": "hot-rows", "6": "sparse"}
tmp = tempfile.mkdtemp()
tus = [f[:-4] for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and not f.endswith("_probe.hip")]
procs = []
for tu in tus:
    out = os.path.join(tmp, tu + ".o")
    procs.append((tu, out, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "--no-gpu-bundle-output",
                                             "-c", "-o", out, os.path.join(CSRC, tu + ".hip")] + sys.argv[2:], stderr=subprocess.DEVNULL)))
floor = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rows = []
for tu, out, pr in procs:
    pr.wait()
    txt = subprocess.run([READELF, "--symbols", "--demangle", "--wide", out], capture_output=True, text=True).stdout
    for line in txt.splitlines():
        m = re.match(r"^\s*\d+:\s+[0-9a-f]+\s+(\d+)\s+FUNC\s+\S+\s+\S+\s+\S+\s+(.*)$", line)
        if m and "kernel" in m.group(2):
            rows.append((int(m.group(1)), tu, m.group(2)))
for size, tu, name in sorted(rows, reverse=True):
    if size >= floor:
        print("%7d  %-22s %s" % (size, tu, name[:150]))
print("%d kernels, %d above 32 KB, %d above 64 KB" % (len(rows), sum(s > 32768 for s, _, _ in rows), sum(s > 65536 for s, _, _ in rows)))
