#!/usr/bin/env python3
"""SURVEY.md s8 f-4 on the one shape where a prefilter's unit of skipping is contiguous text of ONE row: few long rows (the
stripe paths).  The reference skips to `indexOf(prefix)` before it walks the DFA (DFAClassBuilder.java:365-376,
CompilationPolicy.java:44-57).  Here: the Sherlock Holmes text of the reference's own tests replicated to 1 GiB as 1024 rows
of 1 MiB; per literal pattern (a) the product path (containedIn / find on the long rows: speculative stripes for these
automata), (b) the prefix-scan probe kernel alone (needle_amd/csrc/prefix_probe.hip: first 4 bytes of the literal, every
4 KiB stripe, memory speed), (c) the share of stripes a prefix window starts in.  A stripe-level prefilter can at best cost
(b) + (c) x (a); a row-level one (b) + [rows with an occurrence] x (a).  -> gpurun_out/r3/prefix_prefilter_ab.json"""
import ctypes, gzip, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from needle_amd.build import build_probe
from needle_amd.pattern import DFACompiler, unpack_bitmap

text = np.frombuffer(gzip.open(os.path.join(ROOT, "tests", "golden", "sherlockholmes.txt.gz")).read(), dtype=np.uint8)
N_ROWS, ROW = 1024, 1 << 20
reps = (N_ROWS * ROW + len(text) - 1) // len(text)
host = np.tile(text, reps)[:N_ROWS * ROW].reshape(N_ROWS, ROW)
rows = torch.from_numpy(host).cuda()
L = ctypes.CDLL(build_probe())
L.prefix_scan_launch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_int, ctypes.c_void_p]
spr = ROW // 4096
hit = torch.zeros(N_ROWS * spr, dtype=torch.uint8, device="cuda")
first = torch.empty(N_ROWS, dtype=torch.int32, device="cuda")


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        d = time.perf_counter() - t
        best = d if best is None else min(best, d)
    return best * 1e3


out = {"rows": N_ROWS, "row_bytes": ROW, "text": "tests/golden/sherlockholmes.txt.gz x %d" % reps, "patterns": {}}
for regex in ("Sherlock", "Moriarty", "Zanzibar", "Sherlock Holmes|Mycroft", "Adler"):
    lit = regex.split("|")[0].encode()[:4]
    lit32 = int.from_bytes(lit, "little")
    mask = (1 << (8 * len(lit))) - 1
    p = DFACompiler.compile(regex, "t", 0)

    def scan():
        first.fill_(2 ** 31 - 1)
        L.prefix_scan_launch(rows.data_ptr(), N_ROWS, ROW, lit32, mask, hit.data_ptr(), first.data_ptr(), 4096, torch.cuda.current_stream().cuda_stream)
    t_scan = timed(scan)
    f = first.cpu().numpy()
    want = np.array([bytes(host[r]).find(lit) for r in range(0, N_ROWS, 97)])
    got = np.where(f[::97] == 2 ** 31 - 1, -1, f[::97])
    assert (got == want).all(), (regex, got[:5], want[:5])
    hit_frac = float(hit.float().mean().item())
    rows_with = float((f != 2 ** 31 - 1).mean())
    t_c = timed(lambda: p.contained_in_batch(rows))
    t_f = timed(lambda: p.find_batch(rows))
    info = p.info()
    single = "|" not in regex
    out["patterns"][regex] = {
        "prefix": lit.decode(), "kernel_mode_forwards": info["kernel_mode"]["forwards"], "states": info["n_states"]["forwards"],
        "prefix_scan_ms": round(t_scan, 3), "prefix_scan_GBs": round(N_ROWS * ROW / t_scan / 1e6, 1),
        "contained_in_ms": round(t_c, 3), "contained_in_GBs": round(N_ROWS * ROW / t_c / 1e6, 1),
        "find_ms": round(t_f, 3), "find_GBs": round(N_ROWS * ROW / t_f / 1e6, 1),
        "stripes_with_a_prefix_window": round(hit_frac, 4), "rows_with_a_prefix_window": round(rows_with, 4),
        "applicable": single,
        "best_case_stripe_level_ms": round(t_scan + hit_frac * t_c, 3), "best_case_speedup_contained_in": round(t_c / (t_scan + hit_frac * t_c), 2),
        "best_case_row_level_ms": round(t_scan + rows_with * t_c, 3),
    }
    print(regex, out["patterns"][regex])
os.makedirs(os.path.join(ROOT, "gpurun_out", "r3"), exist_ok=True)
os.makedirs(os.path.join(ROOT, "gpurun_out", "r4"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r4", "prefix_prefilter_ab.json"), "w"), indent=1)
