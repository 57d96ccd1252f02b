"""f-2 on the device: a pattern revived from the precompiled-pattern blob (needle_pattern_serialize /
needle_pattern_deserialize: the analogue of Precompile.precompile, NC/precompile/Precompile.java:30-53) runs the kernels
and gives the outputs the reference's own compiled classes gave (tests/golden/snapshots); and a C99 program does the
same round trip through the header alone (tests/c/blob_roundtrip.c)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_snapshot, snapshot_names
from test_gpu_parity import gpu_run, pattern_from_fixture, rows_from_strings


@pytest.mark.gpu
@pytest.mark.parametrize("name", snapshot_names())
def test_deserialized_pattern_equals_reference_bytecode_vectors(name):
    from needle_amd.pattern import Pattern
    doc = load_snapshot(name)
    blob = pattern_from_fixture(doc).to_bytes()
    p = Pattern.from_bytes(blob)  # the original pattern object is gone: only the blob's content reaches the device
    assert p.to_bytes() == blob
    hs = [v["h"] for v in doc["vectors"]]
    rows, lens = rows_from_strings(hs, np.uint16)
    m, c, f, fs, fe = gpu_run(p, rows, lens)
    for i, v in enumerate(doc["vectors"]):
        assert m[i] == v["matches"], ("matches", v["h"])
        assert c[i] == v["containedIn"], ("containedIn", v["h"])
        assert f[i] == v["find"][0], ("find", v["h"])
        assert (fs[i], fe[i]) == ((v["find"][1], v["find"][2]) if v["find"][0] else (-1, -1)), ("start/end", v["h"])


@pytest.mark.gpu
def test_compiled_then_deserialized_patterns_scan_alike():
    """The same for patterns that come out of needle_compile (no snapshot): every kernel mode the BASELINE configs use."""
    import torch
    from needle_amd import workload as W
    from needle_amd.pattern import DFACompiler, Pattern, unpack_bitmap
    words = W.keywords(1000)
    for rx, rows in (("[0-9]+", W.digits_batch(torch, 0, 20000, 256, device="cuda")),
                     ("|".join(words), W.keyword_batch(torch, words, 0, 20000, 256, device="cuda")),
                     (W.script_regex(), W.script_batch(torch, 0, 20000, 256, device="cuda"))):
        p = DFACompiler.compile(rx, "t")
        q = Pattern.from_bytes(p.to_bytes())
        assert q.info() == p.info()
        a, b = p.find_batch(rows), q.find_batch(rows)
        assert all(bool((x == y).all()) for x, y in zip(a, b))
        assert bool((p.contained_in_batch(rows) == q.contained_in_batch(rows)).all())
        assert bool((p.matches_batch(rows) == q.matches_batch(rows)).all())
        assert unpack_bitmap(a[0], 20000).any()


def _build(tmp_path):
    from needle_amd import build
    lib = build.build()
    exe = str(tmp_path / "blob_roundtrip")
    libdir = os.path.dirname(lib)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "blob_roundtrip.c"), "-L", libdir, "-lneedle_hip",
                           "-Wl,-rpath," + libdir, "-o", exe])
    return exe, libdir


def test_c_blob_round_trip_builds_and_runs_without_a_device(tmp_path):
    exe, libdir = _build(tmp_path)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(tmp_path / "p.ndlt")], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "blob round trip ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_blob_round_trip_on_the_device(tmp_path):
    exe, libdir = _build(tmp_path)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(tmp_path / "p.ndlt")], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "device results identical and as expected" in r.stdout, r.stdout + r.stderr
