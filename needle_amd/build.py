"""Builds libneedle_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.  `python -m needle_amd.build`"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libneedle_hip.so")
SOURCES = ["needle_scan_find1.hip", "needle_scan_find2.hip", "needle_scan_contained.hip", "needle_scan_matches.hip", "needle_kernels.hip",
           "needle_stripe.hip", "needle_find_all.hip", "needle_find_all_ls.hip", "needle_compact.hip", "needle_ngram.hip", "needle_ngram_host.cpp", "needle_api.cpp", "needle_tuning.cpp", "needle_multi.cpp", "needle_lower.cpp", "needle_regex.cpp"]
HEADERS = ["needle_device.h", "needle_walk.h", "needle_scan.h", "needle_find_all.h", "needle_lower.h", "needle_regex.h", "needle_ngram.h", "needle_ngram_host.h", "needle_unicode_tables.h", os.path.join("..", "..", "include", "needle_hip.h")]


PROBE_LIB = os.path.join(HERE, "libneedle_probe.so")  # measurement aid for bench.py, not part of the product ABI
PROBE_SRCS = [os.path.join(CSRC, "stream_probe.hip"), os.path.join(CSRC, "prefix_probe.hip"), os.path.join(CSRC, "ngram_probe.hip")]


def build_probe(force=False):
    """Measurement kernels: the trivial read-reduce kernel bench.py uses to measure this GPU's streaming-read ceiling, and the
    literal-prefix scan behind the f-4 A/B (scripts/prefix_prefilter_ab.py)."""
    if force or not os.path.exists(PROBE_LIB) or any(os.path.getmtime(f) > os.path.getmtime(PROBE_LIB) for f in PROBE_SRCS):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-o", PROBE_LIB] + PROBE_SRCS)
    return PROBE_LIB


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def _deps(path, seen=None):
    """`path` and every file it #includes with quotes, transitively (resolved next to the includer, then in include/)."""
    import re
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path, errors="replace") as f:
        for inc in re.findall(r'^\s*#\s*include\s*"([^"]+)"', f.read(), flags=re.M):
            for base in (os.path.dirname(path), os.path.join(HERE, "..", "include")):
                cand = os.path.join(base, inc)
                if os.path.exists(cand):
                    _deps(cand, seen)
                    break
    return seen


def build(force=False, verbose=False):
    build_probe(force)
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-x", "hip", "-Wall", "-Wno-unused-variable"]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in SOURCES:  # one hipcc per translation unit, all at once (the scan kernel's 96 instantiations dominate)
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(f) for f in _deps(os.path.join(CSRC, src))):
            continue  # (the object is newer than its source and every header it includes, transitively)
        cmd = flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
