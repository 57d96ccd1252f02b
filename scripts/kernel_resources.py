#!/usr/bin/env python3
"""VGPRs / scratch of every instantiation of the tiled scan kernel (cross-compiled here, no GPU needed): the kernels run
16 waves per workgroup, i.e. at most 128 VGPRs; anything above spills.  python scripts/kernel_resources.py [--all]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "needle_amd", "csrc")
names = {"0": "matches", "1": "containedIn", "2": "find"}
modes = {"0": "pack", "1": "table8", "2": "table16", "3": "hbm", "4": "pair", "5": "hot-rows", "6": "sparse"}
procs = []
tmp = tempfile.mkdtemp()
for tu in ("needle_scan_matches", "needle_scan_contained", "needle_scan_find1", "needle_scan_find2"):
    out = os.path.join(tmp, tu + ".s")
    procs.append((out, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only",
                                         "-S", "-o", out, os.path.join(CSRC, tu + ".hip")], stderr=subprocess.DEVNULL)))
rows = []
for out, pr in procs:
    pr.wait()
    k = None
    for line in open(out):
        m = re.match(r"\s+\.amdhsa_kernel\s+(\S+)", line)
        if m:
            k = [m.group(1), 0, 0]
            rows.append(k)
        m = re.match(r"\s+\.amdhsa_private_segment_fixed_size\s+(\d+)", line)
        if m and k:
            k[1] = int(m.group(1))
        m = re.match(r"\s+\.amdhsa_next_free_vgpr\s+(\d+)", line)
        if m and k:
            k[2] = int(m.group(1))
print("%d kernels; scratch bytes / VGPRs / kernel" % len(rows))
for k, sc, v in sorted(rows):
    m = re.search(r"ILi(\d)ELi(\d)ELi(\d)ELb(\d)ELi(\d+)E", k)
    if m and (sc > 0 or "--all" in sys.argv):
        print("%4d %4d  %-11s cw%s %-8s %-5s tile %s" % (sc, v, names[m.group(1)], m.group(2), modes[m.group(3)], "guard" if m.group(4) == "1" else "full", m.group(5)))
