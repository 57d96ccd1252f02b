"""Builds libneedle_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.  `python -m needle_amd.build`"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libneedle_hip.so")
SOURCES = ["needle_kernels.hip", "needle_api.cpp", "needle_lower.cpp", "needle_regex.cpp"]
HEADERS = ["needle_device.h", "needle_lower.h", "needle_regex.h", os.path.join("..", "..", "include", "needle_hip.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
           "-Wall", "-Wno-unused-variable", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
