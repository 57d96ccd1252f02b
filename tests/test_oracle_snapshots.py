"""Pins the CPU oracle (oracle/needle_walk.c) against the reference's own compiled output.

The expected values in tests/golden/snapshots/*.json were produced by interpreting the bytecode of the
12 generated classes under needle-compiler/src/test/resources/snapshots/ (generator:
tests/golden/gen_snapshot_vectors.py).  Those classes contain every CPU prefilter the reference emits
(prefix/suffix/infix indexOf, first-byte mask, predicate seek, maxStart, single-char reverse scan,
fixed-length start), so equality here also shows the prefilters are result-transparent on these inputs.
"""
import pytest

from conftest import load_snapshot, snapshot_names
from oracle.walker import OraclePattern, decode_table_strings

import numpy as np


@pytest.mark.parametrize("name", snapshot_names())
def test_table_strings_decode_to_recorded_arrays(name, oracle_lib):
    doc = load_snapshot(name)
    for key, spec in doc["dfas"].items():
        t = decode_table_strings(spec["table_strings"], spec["n_states"], doc["stride"])
        assert t.tolist() == spec["table"], key
        assert not spec["accepts_dead"]


@pytest.mark.parametrize("as_dfa", [False, True], ids=["ref-start-rule", "backward-dfa"])
@pytest.mark.parametrize("name", snapshot_names())
def test_walker_equals_interpreted_bytecode(name, as_dfa, oracle_lib):
    doc = load_snapshot(name)
    pat = OraclePattern.from_fixture(doc, backwards_as_dfa=as_dfa)
    n = 0
    for v in doc["vectors"]:
        h = v["h"]
        assert not isinstance(v["matches"], str) and not isinstance(v["containedIn"], str) and not isinstance(v["find"], str)
        assert pat.matches(h) == v["matches"], ("matches", h)
        assert pat.contained_in(h) == v["containedIn"], ("containedIn", h)
        found, start, end = pat.find(h)
        assert found == v["find"][0], ("find", h)
        assert end == v["find"][2], ("end", h)
        if found:
            assert start == v["find"][1], ("start", h)
            if "find2" in v:  # second find() continues at nextStart = end
                f2, s2, e2 = pat.find(h, start=end)
                assert f2 == v["find2"][0], ("find2", h)
                assert e2 == v["find2"][2], ("end2", h)
                if f2:
                    assert s2 == v["find2"][1], ("start2", h)
        n += 1
    assert n > 200


@pytest.mark.parametrize("name", ["DigitPlus", "UnionOfManyNames", "aDotc"])
def test_batch_drivers_equal_single_calls(name, oracle_lib):
    doc = load_snapshot(name)
    pat = OraclePattern.from_fixture(doc)
    hs = [v["h"] for v in doc["vectors"] if len(v["h"]) <= 48]
    rows = np.zeros((len(hs), 48), dtype=np.uint16)
    lens = np.zeros(len(hs), dtype=np.uint32)
    for i, h in enumerate(hs):
        rows[i, :len(h)] = [ord(c) for c in h]
        lens[i] = len(h)
    m = pat.batch_matches(rows, lens, threads=2)
    c = pat.batch_contained_in(rows, lens, threads=2)
    fm, fs, fe = pat.batch_find(rows, lens, threads=2)
    for i, h in enumerate(hs):
        assert m[i] == pat.matches(h)
        assert c[i] == pat.contained_in(h)
        found, s, e = pat.find(h)
        assert fm[i] == found
        if found:
            assert (fs[i], fe[i]) == (s, e)
        else:
            assert (fs[i], fe[i]) == (-1, -1)
