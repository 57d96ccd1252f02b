"""The automaton behind find-all's "lengths" form (needle_amd/csrc/needle_lower.h: MatchLengths): the forward search
automaton refined until every state remembers ONE pending match length, so that a find() that ends in state s reports
start = end - pend[s] instead of walking the reversed automaton (DFAClassBuilder.java:640-656).  Host-side only: the
refined table is walked here in plain Python the way the kernel walks it (restart at `end` after every match, the row's
end and chars beyond maxChar end the search like a dying transition) and compared, match by match, with the CPU oracle's
repeated find() (oracle/walker.py).  Patterns whose matches have unbounded or ambiguous lengths must be refused."""
import numpy as np
import pytest

from test_compile_matches_txt import oracle_for


def _find_all_by_lengths(ml, class_map, _unused_max_char, text):
    table, acc, pend, nd, max_char = ml["table"], ml["accepting"], ml["pend"], ml["n_dead"], ml["max_char"]
    over = table.shape[1] - 1
    out, cursor, n = [], 0, len(text)
    while True:
        st, last, end_state = 0, -1, 0
        i = cursor
        while True:
            if i >= n:                          # PAD: the row's end, as a dying transition
                nxt = -1
            else:
                nxt = table[st, over if text[i] > max_char else class_map[text[i]]]
            if nxt < 0 or 1 <= nxt <= nd:       # dead, or dead with a match pending
                end_state = nxt if nxt > 0 else (0 if pend[st] == 0 else [k for k in range(1, nd + 1) if pend[k] == pend[st]][0])
                break
            st = nxt
            i += 1
            if acc[st]:
                last = i
        if last < 0:
            return out
        L = int(pend[end_state]) if end_state > 0 else 0
        assert L > 0, (text, cursor, last)
        out.append((last - L, last))
        cursor = last


# (the last rows: alternatives whose ORDER decides what indexBackwards reports -- the reversed automaton is pruned by priority, so
# `bc|abc` finds "bc" inside "abc" -- and cases a GPU fuzz campaign caught when the analysis modelled "the longest match" instead)
KEYWORDS = ["bc|abc", "zzy|xyzzy", "abc|bc", r"(\D|(.|b)([a-cx-z]|1))c", r"A|(\w)?.([a-cx-z]|(B|[x-z0]))", r"[^0-9a-f]|c[a\d]",
            r"(y|((\x41|[a-zA-Z])[^0-9a-f]|(.|a)中|a))",
            "abc|bcd|cdefg|a|xyzzy|zzy", "Sherlock|Holmes|Watson|Irene|Adler|John|Baker", "(ab|a|bcdef|g)", "aab|ab|b", "ab?c|abc?d",
            "http://|https://|ftp|tp:", "[ab]c|a[bc]d|[abc]{4}", "(foo|foobar|bar|barbaz|baz)", "a.c|ab"]


@pytest.mark.parametrize("regex", KEYWORDS)
def test_lengths_automaton_equals_repeated_find(regex, oracle_lib):
    from needle_amd.pattern import DFACompiler
    p = DFACompiler.compile(regex, "t", 0)
    o, _ = oracle_for(regex, 0)
    ml = p.match_length_automaton()
    assert ml is not None, regex
    t = p.tables()
    cm, mc = t["class_map"], t["dfas"]["forwards"]["max_char"]
    alphabet = sorted(set(ord(c) for c in regex if ord(c) < 256 and (c.isalnum() or c in ":/. "))) + [ord(c) for c in " ~_\n019xyzAB"] + [200]
    rng = np.random.default_rng(7)
    for trial in range(400):
        n = int(rng.integers(0, 40))
        text = rng.choice(alphabet, size=n).astype(np.uint8) if n else np.zeros(0, dtype=np.uint8)
        assert _find_all_by_lengths(ml, cm, mc, text.tolist()) == o.find_all(text), (regex, bytes(text))


def test_lengths_automaton_on_the_bench_dictionary(oracle_lib):
    """C3's 1000-keyword union: 1401 states -> 1463 + 3 dead states (lengths 3, 4, 5); the program still fits the LDS."""
    from needle_amd import workload as W
    from needle_amd.pattern import DFACompiler
    words = W.keywords(1000)
    p = DFACompiler.compile("|".join(words), "t", 0)
    ml = p.match_length_automaton()
    assert ml is not None and ml["n_dead"] == 3 and sorted(ml["pend"][1:4].tolist()) == [3, 4, 5]
    assert ml["n_states"] < 1.1 * p.info()["n_states"]["forwards"]
    o, _ = oracle_for("|".join(words), 0)
    t = p.tables()
    rows = W.keyword_batch(np, words, 5, 300, 256)
    for r in rows:
        assert _find_all_by_lengths(ml, t["class_map"], t["dfas"]["forwards"]["max_char"], r.tolist()) == o.find_all(r)


@pytest.mark.parametrize("regex", ["[0-9]+", "a.*b", "(ab)+", "a*", "x?", "[a-z][a-z]+"])
def test_unbounded_or_empty_matches_are_refused(regex):
    from needle_amd.pattern import DFACompiler
    assert DFACompiler.compile(regex, "t", 0).match_length_automaton() is None


@pytest.mark.parametrize("seed", range(3))
def test_lengths_automaton_on_random_regexes(seed, oracle_lib):
    """Seeded random regexes (the generator of tests/test_compile_vs_python_restatement.py) x flag sets: wherever the analysis
    offers a refined automaton, walking it reports exactly the oracle's repeated find() on random haystacks."""
    import random
    from needle_amd.pattern import DFACompiler, PatternException
    from test_compile_vs_python_restatement import FLAG_SETS, random_regex
    rng = random.Random(7000 + seed)
    nrng = np.random.default_rng(seed)
    alphabet = [ord(c) for c in "abcxyz019 AB_\n."] + [0xE9, 0x416, 0x4E2D, 0xFFFF]
    offered = 0
    for _ in range(60):
        regex, flags = random_regex(rng), rng.choice(FLAG_SETS)
        try:
            p = DFACompiler.compile(regex, "t", flags)
            o, _ = oracle_for(regex, flags)
        except (PatternException, ValueError):
            continue
        ml = p.match_length_automaton()
        if ml is None:
            continue
        offered += 1
        t = p.tables()
        for trial in range(150):
            n = int(nrng.integers(0, 40))
            text = nrng.choice(alphabet, size=n).astype(np.uint16)
            assert _find_all_by_lengths(ml, t["class_map"], 0, text.tolist()) == o.find_all(text), (regex, flags, text.tolist())
    assert offered >= 5
