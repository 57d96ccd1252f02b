"""The reference's golden-vector file (RES/matches.txt, transcribed to tests/golden/matches.json) against
  product table generator (needle_compile)  ->  tables  ->  CPU oracle walker (oracle/needle_walk.c).
This is the semantic gate for the generator on regexes that have no compiled snapshot: leftmost-first vs
leftmost-longest, greedy quantifiers, char sets, unicode haystacks, case-insensitivity, dot semantics.
Rows without a flags column run under pseudo-random flags in the reference (DFACompilerTest.java:775-782); here
they run with flags 0 and with every flag set that cannot change the language of that row."""
import json
import os

import pytest

from conftest import GOLDEN

DOC = json.load(open(os.path.join(GOLDEN, "matches.json")))
DOTALL, CI, UCASE, UCC, LML = 0x20, 0x02, 0x40, 0x100, 0x800000


@pytest.fixture(scope="module", autouse=True)
def built():
    from needle_amd import build
    build.build()
    from oracle import walker
    walker.build()


def oracle_for(pattern, flags):
    from needle_amd.pattern import DFACompiler
    from oracle.walker import Dfa, OraclePattern
    t = DFACompiler.compile(pattern, "t", flags).tables()
    d = {k: Dfa(t["class_map"], t["stride"], v["table"], v["accepting"], v["max_char"]) for k, v in t["dfas"].items()}
    return OraclePattern(d["matches"], d["contained_in"], d["forwards"], d["backwards"], t["fixed_len"], -1), t


def flag_sets(row):
    if row["flags"] is not None:
        return [row["flags"]]
    p = row["pattern"]
    out = [0]
    if "." not in p:
        out.append(DOTALL)
    if not any(c.isalpha() for c in p):
        out.append(CI)
    out.append(UCASE)  # does nothing without CASE_INSENSITIVE (Pattern.java:19-21)
    return out


ROWS = [(i, r) for i, r in enumerate(DOC["rows"])]


@pytest.mark.parametrize("i,row", ROWS, ids=["%03d" % i for i, _ in ROWS])
def test_matches_txt_row(i, row):
    # (rows under UNICODE_CASE / UNICODE_CHARACTER_CLASS -- RES/matches.txt:195-200,204-224 -- run on the Unicode 13.0.0
    # Character data of needle_unicode_tables.h: what JDK 15..18 answer)
    for flags in flag_sets(row):
        o, _ = oracle_for(row["pattern"], flags)
        found, start, end = o.find(row["haystack"])
        assert found == row["found"], (row, flags)
        if found:
            assert (start, end) == (row["start"], row["end"]), (row, flags)
        # the algebra SearchMethodTestUtil.java:48-120 ties find to matches/containedIn
        assert o.contained_in(row["haystack"]) == found
        if found:
            assert o.matches(row["haystack"][start:end])


def test_inline_known_answers():
    for case in DOC["inline"]:
        o, _ = oracle_for(case["pattern"], case["flags"])
        h = case["haystack"]
        if "matches" in case:
            assert o.matches(h) == case["matches"], case
        if "find" in case:
            frm = case.get("find_range", [0, len(h)])[0]
            found, s, e = o.find(h, start=frm)
            assert [found, s, e] == case["find"], case


def test_state_counts_per_mode():
    """NFAToDFACompilerTest.java:12-35: (AB){1,2} -> 7 (BASIC... pre-minimisation counts are not observable here);
    what IS observable: the minimised DFAs of the DigitPlus snapshot have 2 states each."""
    _, t = oracle_for("[0-9]+", 0)
    assert [t["dfas"][k]["n_states"] for k in ("matches", "contained_in", "forwards", "backwards")] == [2, 2, 2, 2]
    assert t["stride"] == 4 and t["fixed_len"] == -1


def test_byte_classes_known_answers():
    """DFATest.java:239-250 pins '/' = 2 and ':' = 3 for http://.+ under DOTALL."""
    _, t = oracle_for("http://.+", DOTALL)
    cm = t["class_map"]
    assert cm[ord("/")] == 2 and cm[ord(":")] == 3
    assert all(cm[c] == 1 for c in range(0, ord("/"))) and all(cm[c] == 1 for c in range(ord("0"), ord(":")))
    assert t["stride"] == 8      # 6 groups -> byteClassCount 7 -> rounded up to 8 (DFAClassBuilder.java:240-253)
    _, t0 = oracle_for("http://.+", 0)
    assert t0["stride"] == 16    # + the \n\r group: byteClassCount 8 -> 16
    assert t0["class_map"][10] == t0["class_map"][13] != t0["class_map"][11]


def test_error_conventions():
    from needle_amd.pattern import DFACompiler, PatternSyntaxException
    for bad in ["a{2,1}", "(", "a)", "[a", "*a", "a**?", "\\1", "^a", "a$", "a??", "a{", "\\xZZ", "[z-a]"]:
        with pytest.raises(PatternSyntaxException):
            DFACompiler.compile(bad, "bad", 0)
    with pytest.raises(ValueError):
        DFACompiler.compile("a", "bad", 0x4)  # unknown flag bit (CompilerOptions.java:9-16)
