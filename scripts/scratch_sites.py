#!/usr/bin/env python3
"""Where the scan kernels that have a private segment touch it (cross-compiled here, no GPU): per kernel, scratch loads / stores by loop depth
(0 = outside the loop over 64-row groups, 1 = once per group, >= 2 = inside a group's chunk / piece loops).  python scripts/scratch_sites.py"""
import os, re, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
procs = []
for tu in ("needle_scan_matches", "needle_scan_contained", "needle_scan_find1", "needle_scan_find2"):
    out = os.path.join(tmp, tu + ".s")
    procs.append((out, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S", "-o", out,
                                         os.path.join(ROOT, "needle_amd", "csrc", tu + ".hip")], stderr=subprocess.DEVNULL)))
print("scratch bytes / kernel <OP, CW, MODE, GUARD, tile, LEN> / {depth: (loads, stores)}")
n = 0
for out, pr in procs:
    pr.wait()
    for k in re.split(r"\n\s+\.globl\s+", open(out).read())[1:]:
        m = re.search(r"private_segment_fixed_size\s+(\d+)", k)
        if not m or int(m.group(1)) == 0:
            continue
        name = subprocess.run(["c++filt", k.split("\n")[0].strip()], capture_output=True, text=True).stdout.strip()
        name = name.replace("void needle::scan_kernel", "").split("(")[0]
        cur, stats = 0, {}
        for l in k.split("\n"):
            if re.match(r"^\.LBB", l):
                d = re.search(r"Depth=(\d+)", l)
                cur = int(d.group(1)) if d else 0
            t = l.strip()
            if t.startswith("scratch_"):
                s = stats.setdefault(cur, [0, 0])
                s[0 if "load" in t else 1] += 1
        n += 1
        print("%3s %-34s %s" % (m.group(1), name, {d: tuple(v) for d, v in sorted(stats.items())}))
print(n, "kernels with a private segment")
