"""The reference's own inline asserts (DFACompilerTest.java, transcribed to tests/golden/dfacompilertest_cases.json)
replayed with the helper semantics of SearchMethodTestUtil.java:48-120:

  match(p, s)  : matches, containedIn, find() == (0, len(s)), then find(p, s)
  fail(p, s)   : !containedIn && !matches
  find(p, s[, from, to]) : containedIn; find succeeds; the found substring matches; the prefix before it does not
                 match (unless it is empty and the pattern matches ""); no earlier-starting substring ending at
                 the same end matches; empty-matching patterns anchor at `from`
  find(p, needle, prefix, suffix) under QuickTheories: the same for needle / prefix+needle / prefix+needle+suffix /
                 needle+suffix with seeded noise over A-Z and U+00C5..U+00C9 (ALPHABET, DFACompilerTest.java:28-30)

CPU: product table generator -> tables -> oracle walker.  GPU (-m gpu): the same cases through the Matcher mirror."""
import json
import os
import random

import pytest

from conftest import GOLDEN
from test_compile_matches_txt import oracle_for

DOC = json.load(open(os.path.join(GOLDEN, "dfacompilertest_cases.json")))
CASES = DOC["cases"]


class OracleFacade:
    """Adapts the oracle to the few Matcher calls the helpers need."""

    def __init__(self, regex, flags):
        self.o, _ = oracle_for(regex, flags)

    def matches(self, s):
        return self.o.matches(s)

    def contained_in(self, s):
        return self.o.contained_in(s)

    def find(self, s, frm=0):
        return self.o.find(s, start=frm)


class GpuFacade:
    def __init__(self, regex, flags):
        from needle_amd.pattern import DFACompiler
        self.p = DFACompiler.compile(regex, "t", flags)

    def matches(self, s):
        return self.p.matcher(s).matches()

    def contained_in(self, s):
        return self.p.matcher(s).containedIn()

    def find(self, s, frm=0):
        m = self.p.matcher(s)
        f = m.find(frm, len(s))
        return (True, m.start(), m.end()) if f else (False, None, m.end())


def helper_find(p, s, frm=0):
    assert p.contained_in(s), ("containedIn", s)
    found, start, end = p.find(s, frm)
    assert found, ("find", s, frm)
    assert p.matches(s[start:end]), ("found substring must match", s, start, end)
    prefix = s[frm:start]
    assert (p.matches("") and prefix == "") or not p.matches(prefix), ("prefix condition", s, prefix)
    for k in range(frm, start):
        assert not p.matches(s[k:end]), ("earlier start matches", s, k, end)
    if p.matches(""):
        assert start == frm


def helper_match(p, s):
    assert p.matches(s) and p.contained_in(s)
    assert p.find(s) == (True, 0, len(s)), (s, p.find(s))
    helper_find(p, s)


def helper_fail(p, s):
    assert not p.contained_in(s) and not p.matches(s), s


def noise(rng):
    if rng.random() < 0.5:
        return "".join(chr(rng.randint(65, 90)) for _ in range(rng.randint(0, 10)))
    return "".join(chr(rng.randint(0xC5, 0xC9)) for _ in range(rng.randint(0, 10)))


def run_case(p, case, rng, rounds):
    op, s = case["op"], case["s"]
    if op == "match":
        helper_match(p, s)
    elif op == "fail":
        helper_fail(p, s)
    elif op == "find":
        helper_find(p, s, case.get("range", [0])[0])
    elif op == "find_noise":
        for _ in range(rounds):
            pre, suf = noise(rng), noise(rng)
            for h in (s, pre + s, pre + s + suf, s + suf):
                helper_find(p, h)
    elif op == "assert_matches":
        assert p.matches(s) == case["expect"], case
    elif op == "assert_containedIn":
        assert p.contained_in(s) == case["expect"], case
    else:
        raise AssertionError(op)


def grouped():
    by = {}
    for c in CASES:
        by.setdefault((c["regex"], c["flags"]), []).append(c)
    return sorted(by.items(), key=lambda kv: kv[0])


@pytest.mark.parametrize("key,cases", grouped(), ids=[repr(k[0])[:30] for k, _ in grouped()])
def test_reference_inline_asserts_oracle(key, cases, oracle_lib):
    from needle_amd import build
    build.build()
    p = OracleFacade(*key)
    rng = random.Random(hash(key[0]) & 0xFFFF)
    for c in cases:
        run_case(p, c, rng, rounds=25)


@pytest.mark.gpu
def test_reference_inline_asserts_gpu_matcher():
    n = 0
    for key, cases in grouped():
        p = GpuFacade(*key)
        rng = random.Random(7)
        for c in cases:
            run_case(p, c, rng, rounds=1)
            n += 1
    assert n > 100


# DFACompilerTest.java:623-633,663-699: sherlockStreetInFile / upperOrLowercaseSherlockInFile walk every
# non-overlapping match of the regex over `Files.readAllLines(sherlockholmes.txt).get(0)` with repeated find() and
# require the JDK's (start, end) each time.  That first line (it begins with U+FEFF, so it takes the UTF-16 path):
SHERLOCK_LINE0 = "﻿The Project Gutenberg eBook of The Adventures of Sherlock Holmes, by Arthur Conan Doyle"
FILE_CASES = [("Sherlock|Street", [(50, 58)]), ("[Ss]herlock", [(50, 58)]),
              # findingIngWords, :603-614: two matches by repeated find()
              ("[a-zA-Z]+ing", None)]
ING_HAYSTACK = "the most perfect reasoning and observing machine that the world has seen"


def stdlib_matches(regex, s):
    import re
    return [(m.start(), m.end()) for m in re.finditer(regex, s)]


def test_file_based_repeated_find_oracle(oracle_lib):
    for regex, want in FILE_CASES:
        s = SHERLOCK_LINE0 if want is not None else ING_HAYSTACK
        want = want if want is not None else stdlib_matches(regex, s)
        o, _ = oracle_for(regex, 0)
        assert o.find_all(s) == want
    assert len(stdlib_matches("[a-zA-Z]+ing", ING_HAYSTACK)) == 2


@pytest.mark.gpu
def test_file_based_repeated_find_gpu_matcher():
    from needle_amd.pattern import DFACompiler
    for regex, want in FILE_CASES:
        s = SHERLOCK_LINE0 if want is not None else ING_HAYSTACK
        want = want if want is not None else stdlib_matches(regex, s)
        m = DFACompiler.compile(regex, "fileSearchRegex").matcher(s)
        got = []
        while m.find():
            got.append((m.start(), m.end()))
        assert got == want, regex
