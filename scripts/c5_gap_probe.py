#!/usr/bin/env python3
"""C5 `find` is 0.87 ms per launch on some boxes / runs and 0.94 on others ("bimodal and nobody has said why").  This times the SAME
kernel on the SAME resident batch three ways in one process: launches back to back (what bench.py's timed steps do), launches with the device
left idle for a few ms between them, and back to back again -- each with HIP events around every launch, clocks and power from rocm-smi between
the phases.  If the gapped launches are faster, the slow mode is the device's power management under sustained load, not the kernel.
Usage: python scripts/c5_gap_probe.py [workload] [launches]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

w = sys.argv[1] if len(sys.argv) > 1 else "c5"
n_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 60
pattern, label, words = bench.make_pattern(w)
rows = bench.make_rows(w, words, 0, 10_000_000, "cuda:0")
op = pattern.contained_in_batch if w == "c2" else pattern.find_batch
out = op(rows)
torch.cuda.synchronize()


def smi():
    try:
        t = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [ln.split(":", 1)[1].strip() for ln in t.splitlines() if any(k in ln for k in ("sclk", "mclk", "fclk", "Power", "junction"))]
        return " | ".join(keep)[:300]
    except Exception as e:  # noqa: BLE001
        return "rocm-smi unavailable: %s" % e


def phase(name, gap_s):
    evs = []
    for _ in range(n_launch):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        op(rows, out=out)
        b.record()
        evs.append((a, b))
        if gap_s:
            torch.cuda.synchronize()
            time.sleep(gap_s)
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    print("%-34s kernel ms: min %.4f median %.4f max %.4f   %s" % (name, ms[0], ms[len(ms) // 2], ms[-1], smi()))


print(w, label, "--", smi())
scratch = torch.empty_like(rows)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15:  # bench.py's pre-warm
    scratch.copy_(rows)
    torch.cuda.synchronize()
del scratch
phase("back to back", 0)
phase("5 ms idle between launches", 0.005)
phase("back to back again", 0)
phase("1 ms idle between launches", 0.001)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.0:  # two seconds of sustained load, then once more
    for _ in range(50):
        op(rows, out=out)
    torch.cuda.synchronize()
phase("back to back after 2 s of load", 0)
