"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
  (1) the outputs recorded from the reference's own compiled classes (tests/golden/snapshots), and
  (2) the CPU oracle (oracle/needle_walk.c) on seeded batches, incl. ragged / empty / 8- and 16-bit rows.
Bit-exact: matches/containedIn bitmaps and find (matched, start, end)."""
import os

import numpy as np
import pytest

from conftest import load_snapshot, snapshot_names

NAMES = {"matches": "Matches", "contained_in": "ContainedIn", "forwards": "Forwards", "backwards": "Backwards"}


def pattern_from_fixture(doc, backwards_as_dfa=True):
    from needle_amd.pattern import Pattern
    from oracle.walker import class_map_from_runs
    cm = class_map_from_runs(doc["class_map_runs"])
    dfas = {k: dict(n_states=doc["dfas"][v]["n_states"], max_char=doc["dfas"][v]["max_char"],
                    accepting=doc["dfas"][v]["accepting"], table_strings=doc["dfas"][v]["table_strings"])
            for k, v in NAMES.items()}
    bk = doc["backwards"]
    fixed = bk["len"] if bk["kind"] == "fixed_len" and not backwards_as_dfa else -1
    return Pattern.from_tables(cm, doc["stride"], dfas, fixed_len=fixed)


def rows_from_strings(hs, dtype, stride=None):
    stride = stride or max(16 // np.dtype(dtype).itemsize, max((len(h) for h in hs), default=1))
    rows = np.zeros((len(hs), stride), dtype=dtype)
    lens = np.zeros(len(hs), dtype=np.uint32)
    for i, h in enumerate(hs):
        rows[i, :len(h)] = [ord(c) for c in h]
        lens[i] = len(h)
    return rows, lens


def gpu_run(p, rows, lens):
    import torch
    from needle_amd.pattern import unpack_bitmap
    n = rows.shape[0]
    pad = (-rows.shape[1] * rows.dtype.itemsize) % 16 // rows.dtype.itemsize
    if pad:
        rows = np.concatenate([rows, np.zeros((n, pad), dtype=rows.dtype)], axis=1)
    t = torch.from_numpy(rows.view(np.int16) if rows.dtype == np.uint16 else rows).cuda()
    tl = None if lens is None else torch.from_numpy(lens.astype(np.int32)).cuda()
    m = unpack_bitmap(p.matches_batch(t, tl), n)
    c = unpack_bitmap(p.contained_in_batch(t, tl), n)
    fw, fs, fe = p.find_batch(t, tl)
    torch.cuda.synchronize()
    return m, c, unpack_bitmap(fw, n), fs.cpu().numpy(), fe.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("fixed", [False, True], ids=["backward-dfa", "ref-start-rule"])
@pytest.mark.parametrize("name", snapshot_names())
def test_gpu_equals_reference_bytecode_vectors(name, fixed):
    doc = load_snapshot(name)
    p = pattern_from_fixture(doc, backwards_as_dfa=not fixed)
    hs = [v["h"] for v in doc["vectors"]]
    rows, lens = rows_from_strings(hs, np.uint16)
    m, c, f, fs, fe = gpu_run(p, rows, lens)
    for i, v in enumerate(doc["vectors"]):
        assert m[i] == v["matches"], ("matches", v["h"])
        assert c[i] == v["containedIn"], ("containedIn", v["h"])
        assert f[i] == v["find"][0], ("find", v["h"])
        if v["find"][0]:
            assert (fs[i], fe[i]) == (v["find"][1], v["find"][2]), ("start/end", v["h"], fs[i], fe[i])
        else:
            assert (fs[i], fe[i]) == (-1, -1)
    # the 8-bit path on the Latin-1 subset of the same vectors
    idx = [i for i, h in enumerate(hs) if all(ord(ch) < 256 for ch in h)]
    rows8, lens8 = rows_from_strings([hs[i] for i in idx], np.uint8)
    m8, c8, f8, fs8, fe8 = gpu_run(p, rows8, lens8)
    assert (m8 == m[idx]).all() and (c8 == c[idx]).all() and (f8 == f[idx]).all()
    assert (fs8 == fs[idx]).all() and (fe8 == fe[idx]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["DigitPlus", "UnionOfManyNames", "HolmesNearWatson", "aDotc", "RepeatingUnionOfShortStrings"])
@pytest.mark.parametrize("n_rows,stride,ragged", [(1, 256, False), (63, 256, False), (4096 + 17, 256, False),
                                                   (3000, 256, True), (1000, 48, True), (777, 384, False), (0, 256, False)])
def test_gpu_equals_oracle_on_seeded_batches(name, n_rows, stride, ragged, oracle_lib):
    from oracle.walker import OraclePattern
    doc = load_snapshot(name)
    p = pattern_from_fixture(doc)
    o = OraclePattern.from_fixture(doc, backwards_as_dfa=True)
    rng = np.random.default_rng(1234 + n_rows + stride)
    alpha = sorted(set(ord(c) for c in doc["regex"] if c.isalnum())) + [32, 46, 10, 120, 48, 57]
    rows = rng.choice(np.array(alpha, dtype=np.uint8), size=(n_rows, stride)).astype(np.uint8)
    seeds = [v["h"] for v in doc["vectors"] if 0 < len(v["h"]) <= 32 and all(ord(ch) < 128 for ch in v["h"])]
    for r in range(0, n_rows, 3):  # plant reference haystacks so matches occur
        s = seeds[rng.integers(len(seeds))]
        pos = rng.integers(0, stride - len(s) + 1)
        rows[r, pos:pos + len(s)] = [ord(ch) for ch in s]
    lens = rng.integers(0, stride + 1, size=n_rows).astype(np.uint32) if ragged else None
    if n_rows == 0:
        import torch
        t = torch.zeros((0, stride), dtype=torch.uint8, device="cuda")
        assert p.matches_batch(t).numel() == 0
        return
    m, c, f, fs, fe = gpu_run(p, rows, lens)
    om = o.batch_matches(rows, lens)
    oc = o.batch_contained_in(rows, lens)
    of, ofs, ofe = o.batch_find(rows, lens)
    assert (m == om).all()
    assert (c == oc).all()
    assert (f == of).all() and (fs == ofs).all() and (fe == ofe).all()
    if not ragged and n_rows > 100:
        assert oc.any() and om.sum() < n_rows
    # same batch as UTF-16 code units with some non-Latin-1 chars mixed in
    rows16 = rows.astype(np.uint16)
    mask = rng.random(rows16.shape) < 0.02
    rows16[mask] = rng.choice(np.array([0x3b5, 0x3bb, 0xFFFF, 0x4e2d, 0x100], dtype=np.uint16), size=int(mask.sum()))
    m, c, f, fs, fe = gpu_run(p, rows16, lens)
    assert (m == o.batch_matches(rows16, lens)).all()
    assert (c == o.batch_contained_in(rows16, lens)).all()
    of, ofs, ofe = o.batch_find(rows16, lens)
    assert (f == of).all() and (fs == ofs).all() and (fe == ofe).all()


@pytest.mark.gpu
def test_matcher_mirror_single_strings():
    """Reference Matcher semantics incl. the nextStart cursor (DFACompilerTest.java:66-78,815-825)."""
    doc = load_snapshot("DigitPlus")
    p = pattern_from_fixture(doc)
    m = p.matcher("ab12cd345")
    assert m.find() and (m.start(), m.end()) == (2, 4)
    assert m.find() and (m.start(), m.end()) == (6, 9)
    assert not m.find()
    assert not m.find()
    assert p.matcher("12345").matches() and not p.matcher("1234a").matches()
    assert p.matcher("xx9").containedIn() and not p.matcher("").containedIn()
    m = p.matcher("12 34")
    assert m.find(3, 5) and (m.start(), m.end()) == (3, 5)


@pytest.mark.gpu
def test_zero_length_rows():
    """row_len == 0 (and the Matcher mirror on ""): the verdict is the start state's, no char is consumed."""
    import torch
    from needle_amd.pattern import DFACompiler, unpack_bitmap
    star = DFACompiler.compile("[0-9A-Za-z]*", "s", 0)
    plus = DFACompiler.compile("[0-9]+", "p", 0)
    assert star.matcher("").matches() and star.matcher("").containedIn() and star.matcher("").find()
    assert not plus.matcher("").matches() and not plus.matcher("").containedIn() and not plus.matcher("").find()
    rows = torch.zeros((130, 16), dtype=torch.uint8, device="cuda")
    lens = torch.zeros(130, dtype=torch.int32, device="cuda")
    assert unpack_bitmap(star.matches_batch(rows, lens), 130).all()
    assert not unpack_bitmap(plus.matches_batch(rows, lens), 130).any()
    fw, fs, fe = star.find_batch(rows, lens)
    assert unpack_bitmap(fw, 130).all() and (fs == 0).all() and (fe == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n_rows", [1, 65, 70, 129, 1000])
@pytest.mark.parametrize("stride", [16, 32, 48])
def test_narrow_rows_near_the_buffer_end(n_rows, stride, oracle_lib):
    """Rows narrower than a 128-byte tile chunk: the unclamped fast path must leave the trailing groups to the
    clamped tail (reads may not cross the end of the caller's buffer) and results stay bit-exact."""
    from oracle.walker import OraclePattern
    doc = load_snapshot("DigitPlus")
    p = pattern_from_fixture(doc)
    o = OraclePattern.from_fixture(doc, backwards_as_dfa=True)
    rng = np.random.default_rng(n_rows * 131 + stride)
    rows = rng.choice(np.frombuffer(b"ab 0123456789xyz", dtype=np.uint8), size=(n_rows, stride)).astype(np.uint8)
    lens = rng.integers(0, stride + 1, size=n_rows).astype(np.uint32)
    for l in (None, lens):
        m, c, f, fs, fe = gpu_run(p, rows, l)
        assert (m == o.batch_matches(rows, l)).all() and (c == o.batch_contained_in(rows, l)).all()
        of, ofs, ofe = o.batch_find(rows, l)
        assert (f == of).all() and (fs == ofs).all() and (fe == ofe).all()


@pytest.mark.gpu
def test_single_char_table_mode_stays_covered():
    """Mid-size automata default to the pair table (two chars per lookup); the one-char uint8 table mode they fall back
    to when the pair table does not fit is re-run here on the reference's bytecode vectors with the pair mode disabled
    (the switch is read once per process, hence the subprocess)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NEEDLE_PAIR_MAX_BYTES="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-q", "-x",
                        "-k", "bytecode_vectors or seeded_batches"], cwd=root, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
