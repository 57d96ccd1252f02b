"""Real text: the reference's own test data file (needle-compiler/src/test/resources/sherlockholmes.txt, committed
gzipped as tests/golden/sherlockholmes.txt.gz; the reference's DFACompilerTest reads its first line, :623-633).
Its 12 305 lines are one ragged UTF-16 batch; the whole text is one long row.  All the snapshot regexes
(SnapshotTests.java:30-57) and a few everyday ones run over both: GPU == oracle, and == Python's `re` for the regexes
whose leftmost-first semantics coincide with it."""
import gzip
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, load_snapshot, snapshot_names
from test_gpu_configs import compiled


def text():
    with gzip.open(os.path.join(GOLDEN, "sherlockholmes.txt.gz"), "rb") as f:
        return f.read().decode("utf-8")


REGEXES = sorted({load_snapshot(n)["regex"] for n in snapshot_names()} | {"[a-zA-Z]+ing", "http://.+", "[A-Z][a-z]+ [A-Z][a-z]+", "\\d{4}"})
SAME_AS_RE = ["Sherlock|Street", "[Ss]herlock", "[a-zA-Z]+ing", "[0-9]+", "Sherlock", "\\d{4}", "[A-Z][a-z]+ [A-Z][a-z]+"]


@pytest.mark.gpu
def test_every_line_of_the_text():
    from needle_amd.pattern import pack_strings, unpack_bitmap
    lines = text().split("\n")
    assert len(lines) > 12000
    data, offsets = pack_strings(lines)
    n = len(lines)
    width = max(len(l) for l in lines)
    pad = np.zeros((n, width), dtype=np.uint16)
    lens = np.zeros(n, dtype=np.uint32)
    for i, l in enumerate(lines):
        u = np.frombuffer(l.encode("utf-16-le", "surrogatepass"), dtype=np.uint16)
        pad[i, :u.size] = u
        lens[i] = u.size
    for regex in REGEXES:
        p, o = compiled(regex)
        fw, fs, fe = p.find_packed(data, offsets)
        of, os_, oe = o.batch_find(pad, lens, threads=8)
        got = unpack_bitmap(fw, n)
        assert (got == of).all() and (fs == os_).all() and (fe == oe).all(), regex
        assert (unpack_bitmap(p.contained_in_packed(data, offsets), n) == o.batch_contained_in(pad, lens, threads=8)).all(), regex
        assert (unpack_bitmap(p.matches_packed(data, offsets), n) == o.batch_matches(pad, lens, threads=8)).all(), regex
        if regex in SAME_AS_RE:
            rx = re.compile(regex)
            for i in range(0, n, 7):
                m = rx.search(lines[i])
                assert bool(got[i]) == (m is not None), (regex, i)
                if m:
                    assert (fs[i], fe[i]) == (m.start(), m.end()), (regex, i)


@pytest.mark.gpu
def test_the_whole_text_as_one_row():
    import torch
    from needle_amd.pattern import unpack_bitmap
    t = text()
    units = np.frombuffer(t.encode("utf-16-le", "surrogatepass"), dtype=np.uint16).copy()
    pad = np.zeros((1, (units.size + 7) // 8 * 8), dtype=np.uint16)
    pad[0, :units.size] = units
    rows = torch.from_numpy(pad.view(np.int16)).cuda()
    lens = torch.tensor([units.size], dtype=torch.int32, device="cuda")
    L = np.array([units.size], dtype=np.uint32)
    for regex in ("[0-9]+", "Sherlock", "a.c", "ε|λ", "Sherlock|Street", "[a-zA-Z]+ing"):
        p, o = compiled(regex)
        fw, fs, fe = p.find_batch(rows, lens)
        of, os_, oe = o.batch_find(pad, L, threads=1)
        assert bool(unpack_bitmap(fw, 1)[0]) == bool(of[0]) and (int(fs[0]), int(fe[0])) == (int(os_[0]), int(oe[0])), regex
        m = re.search(regex, t)
        assert (m is not None) == bool(of[0]) and (m is None or (m.start(), m.end()) == (int(os_[0]), int(oe[0]))), regex
        assert bool(unpack_bitmap(p.contained_in_batch(rows, lens), 1)[0]) == bool(of[0]), regex
        assert not unpack_bitmap(p.matches_batch(rows, lens), 1)[0], regex
