"""The C++ table generator (needle_compile, needle_amd/csrc/needle_regex.cpp) against the tables the reference
baked into its 12 committed snapshot classes (regexes: SnapshotTests.java:30-57, flags 0): char-class map, row
stride, all four transition tables WITH the reference's state numbering, accepting sets, maxChar constants and
the fixed-length rule.  CPU only (no kernels run)."""
import numpy as np
import pytest

from conftest import load_snapshot, snapshot_names
from oracle.walker import class_map_from_runs

KEYS = {"matches": "Matches", "contained_in": "ContainedIn", "forwards": "Forwards", "backwards": "Backwards"}


@pytest.fixture(scope="module", autouse=True)
def built():
    from needle_amd import build
    build.build()


@pytest.mark.parametrize("name", snapshot_names())
def test_compiled_tables_equal_snapshot_tables(name):
    from needle_amd.pattern import DFACompiler
    doc = load_snapshot(name)
    t = DFACompiler.compile(doc["regex"], name, doc["flags"]).tables()
    assert t["stride"] == doc["stride"]
    assert (t["class_map"] == class_map_from_runs(doc["class_map_runs"])).all()
    for k, v in KEYS.items():
        spec = doc["dfas"][v]
        got = t["dfas"][k]
        assert got["n_states"] == spec["n_states"], k
        assert got["accepting"] == spec["accepting"], k
        assert got["table"].tolist() == spec["table"], k
        if spec["max_char"] is not None:
            assert got["max_char"] == spec["max_char"], k
        elif k != "backwards" or doc["backwards"]["kind"] == "dfa":
            # no check emitted <=> maxChar == Character.MAX_VALUE (the class has no table-driven
            # indexBackwards at all when find() uses the fixed-length or single-char rule)
            assert got["max_char"] == 0xFFFF, k
    bk = doc["backwards"]
    assert t["fixed_len"] == (bk["len"] if bk["kind"] == "fixed_len" else -1)
