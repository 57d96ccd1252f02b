"""needle_compile (C++: needle_amd/csrc/needle_regex.cpp) against the independent Python restatement of the
reference's compile pipeline (oracle/needle_compile.py), tables bit-for-bit with the reference's state numbering.

1. The Python restatement is first pinned to the reference itself: it must reproduce the tables baked into the 12
   committed snapshot classes (tests/golden/snapshots, regexes from SnapshotTests.java:30-57).
2. Then both generators run over seeded random regexes x flag sets; tables, class maps, strides, maxChar, the
   fixed-length rule and the error class (syntax / unsupported) must agree.
CPU only (no kernels run)."""
import random

import pytest

from conftest import load_snapshot, snapshot_names
from oracle import needle_compile as nc
from oracle.walker import class_map_from_runs

KEYS = {"matches": "Matches", "contained_in": "ContainedIn", "forwards": "Forwards", "backwards": "Backwards"}
FLAG_SETS = [0, nc.DOTALL, nc.CASE_INSENSITIVE, nc.LEFTMOST_LONGEST, nc.CASE_INSENSITIVE | nc.UNICODE_CASE,
             nc.DOTALL | nc.LEFTMOST_LONGEST, nc.UNICODE_CHARACTER_CLASS, nc.CASE_INSENSITIVE | nc.UNICODE_CHARACTER_CLASS]


@pytest.fixture(scope="module", autouse=True)
def built():
    from needle_amd import build
    build.build()


@pytest.mark.parametrize("name", snapshot_names())
def test_python_restatement_reproduces_snapshot_tables(name):
    doc = load_snapshot(name)
    t = nc.compile_regex(doc["regex"], doc["flags"])
    assert t["stride"] == doc["stride"]
    assert t["class_map"] == class_map_from_runs(doc["class_map_runs"]).tolist()[:65536]
    for k, v in KEYS.items():
        spec, got = doc["dfas"][v], t["dfas"][k]
        assert got["n_states"] == spec["n_states"], k
        assert got["accepting"] == spec["accepting"], k
        assert got["table"] == spec["table"], k
        if spec["max_char"] is not None:
            assert got["max_char"] == spec["max_char"], k
    bk = doc["backwards"]
    assert t["fixed_len"] == (bk["len"] if bk["kind"] == "fixed_len" else -1)


ATOMS = "abcxyz019 AB_"
SETS = ["[a-c]", "[^a]", "[0-9]", "[a-zA-Z]", "[ab-]", "[^0-9a-f]", "[x-z0]", "[a-cx-z]", "[[a-b]c]", "[\\]a]", "[a\\d]"]
ESCAPES = ["\\d", "\\w", "\\s", "\\D", "\\W", "\\S", "\\h", "\\v", "\\x41", "\\t", "\\.", "\\0101", "\\H", "\\V", "\\e"]
BROKEN = ["(", ")", "[a", "a{2", "a{3,1}", "*a", "a**?", "\\", "\\p", "a|", "|a", "^a", "a$", "\\1", "a*?", "a++", "[b-a]"]


def random_regex(rng, depth=0):
    r = rng.random()
    if depth > 3 or r < 0.3:
        k = rng.random()
        if k < 0.5:
            return rng.choice(ATOMS)
        if k < 0.6:
            return "."
        if k < 0.75:
            return rng.choice(SETS)
        if k < 0.9:
            return rng.choice(ESCAPES)
        return rng.choice(["é", "Ж", "中", "￿", "µ", "ß", "ẞ", "K", "ǅ", "[α-γ]", "İ"])
    if r < 0.55:
        return random_regex(rng, depth + 1) + random_regex(rng, depth + 1)
    if r < 0.7:
        return "(" + random_regex(rng, depth + 1) + "|" + random_regex(rng, depth + 1) + ")"
    if r < 0.8:
        return random_regex(rng, depth + 1) + "|" + random_regex(rng, depth + 1)
    if r < 0.87:
        return "(" + random_regex(rng, depth + 1) + ")" + rng.choice("*+?")
    if r < 0.93:
        return "(?:" + random_regex(rng, depth + 1) + ")" + rng.choice(["{2}", "{1,3}", "{0,2}", "{0}"])
    return random_regex(rng, depth + 1) + rng.choice("*+?")


def both(regex, flags):
    from needle_amd.pattern import DFACompiler, PatternSyntaxException, PatternClassCompilationException
    try:
        py, py_err = nc.compile_regex(regex, flags), None
    except nc.PatternSyntaxError:
        py, py_err = None, "syntax"
    except nc.Unsupported:
        py, py_err = None, "unsupported"
    try:
        cc, cc_err = DFACompiler.compile(regex, "Fuzz", flags).tables(), None
    except PatternSyntaxException:
        cc, cc_err = None, "syntax"
    except PatternClassCompilationException:
        cc, cc_err = None, "unsupported"
    return py, py_err, cc, cc_err


def assert_same(regex, flags):
    py, py_err, cc, cc_err = both(regex, flags)
    assert py_err == cc_err, (regex, flags)
    if py is None:
        return False
    assert py["stride"] == cc["stride"], (regex, flags)
    assert py["class_map"] == cc["class_map"].tolist()[:65536], (regex, flags)
    assert py["fixed_len"] == cc["fixed_len"], (regex, flags)
    for k, a in py["dfas"].items():
        b = cc["dfas"][k]
        assert a["n_states"] == b["n_states"], (regex, flags, k)
        assert a["accepting"] == b["accepting"], (regex, flags, k)
        assert a["table"] == b["table"].tolist(), (regex, flags, k)
        assert a["max_char"] == b["max_char"], (regex, flags, k)
    return True


@pytest.mark.parametrize("seed", range(6))
def test_random_regexes_compile_to_identical_tables(seed):
    rng = random.Random(1000 + seed)
    compiled = sum(assert_same(random_regex(rng), rng.choice(FLAG_SETS)) for _ in range(60))
    assert compiled > 30


@pytest.mark.parametrize("regex", BROKEN)
def test_rejected_regexes_are_rejected_alike(regex):
    for flags in (0, nc.CASE_INSENSITIVE):
        py, py_err, cc, cc_err = both(regex, flags)
        assert py_err == cc_err, (regex, py_err, cc_err)


UNICODE_KNOWN = [
    # (regex, flags, haystack, matches()) -- facts of the Unicode standard behind java.lang.Character
    ("µ", nc.CASE_INSENSITIVE | nc.UNICODE_CASE, "\u039c", True),   # MICRO SIGN folds with GREEK MU
    ("µ", nc.CASE_INSENSITIVE | nc.UNICODE_CASE, "\u03bc", True),
    ("µ", nc.CASE_INSENSITIVE, "\u03bc", False),                    # ASCII-only folding without UNICODE_CASE
    ("k", nc.CASE_INSENSITIVE | nc.UNICODE_CASE, "\u212a", True),   # KELVIN SIGN
    ("s", nc.CASE_INSENSITIVE | nc.UNICODE_CASE, "\u017f", True),   # LONG S
    ("i", nc.CASE_INSENSITIVE | nc.UNICODE_CASE, "\u0130", True),   # dotted capital I: toLowerCase = i
    ("ß", nc.CASE_INSENSITIVE | nc.UNICODE_CASE, "\u1e9e", False),  # toUpperCase(ß) = ß: not cased for the reference
    ("\u1e9e", nc.CASE_INSENSITIVE | nc.UNICODE_CASE, "ß", True),   # ... but toLowerCase(U+1E9E) = ß
    ("[а-в]", nc.CASE_INSENSITIVE | nc.UNICODE_CASE, "Б", True),     # Cyrillic range
    ("ǆ", nc.CASE_INSENSITIVE | nc.UNICODE_CHARACTER_CLASS, "ǅ", True),  # titlecase digraph (UCC implies UNICODE_CASE)
    ("\\d+", nc.UNICODE_CHARACTER_CLASS, "\u0663\uff15", True),    # ARABIC-INDIC / FULLWIDTH digits
    ("\\d", nc.UNICODE_CHARACTER_CLASS, "\u00b2", False),          # SUPERSCRIPT TWO is No, not Nd
    ("\\d", 0, "\u0663", False),
    ("\\w+", nc.UNICODE_CHARACTER_CLASS, "h\u00e9llo_\u0661\u0301\u2160\u24b6", True),  # letters, Pc, Nd, Mn, Nl, Other_Alphabetic
    ("\\w", nc.UNICODE_CHARACTER_CLASS, "-", False),
    ("\\W", nc.UNICODE_CHARACTER_CLASS, "\u00e9", False),
    ("\\s", nc.UNICODE_CHARACTER_CLASS, "\u2028", True),
    ("\\s", nc.UNICODE_CHARACTER_CLASS, "\u00a0", False),          # NO-BREAK SPACE is not Character.isWhitespace
    ("\\s", nc.UNICODE_CHARACTER_CLASS, "\u001f", True),
    ("\\S", nc.UNICODE_CHARACTER_CLASS, "\u3000", False),
]


@pytest.mark.parametrize("regex,flags,h,want", UNICODE_KNOWN)
def test_unicode_flags_known_answers(regex, flags, h, want, oracle_lib):
    """UNICODE_CASE / UNICODE_CHARACTER_CLASS (RegexParser.java:40-63,212-247,277-291) on the Unicode 13.0.0 data of
    needle_unicode_tables.h: both generators agree table for table, and the oracle walker gives the known answer."""
    from oracle.walker import Dfa, OraclePattern
    assert assert_same(regex, flags)
    t = nc.compile_regex(regex, flags)
    d = {k: Dfa(t["class_map"][:65536], t["stride"], v["table"], v["accepting"], v["max_char"]) for k, v in t["dfas"].items()}
    o = OraclePattern(d["matches"], d["contained_in"], d["forwards"], d["backwards"], t["fixed_len"], -1)
    assert o.matches(h) == want, (regex, flags, h)
