"""SURVEY.md s8 f-4, host side: the n-gram candidate filter's ANALYSIS (needle_amd/csrc/needle_ngram_host.cpp) -- read off the
lowered table: the run-up K after which a restarted walk has caught up, the shortest match, the 4-byte windows that can stand
ahead of a first accepting transition -- checked by running the filter ALGORITHM in plain Python (tests/prefilter_sim.py) on
the reference-layout tables against the CPU oracle: with every window counted as a candidate (is a restart K chars ahead
exact?) and with the real bitmap (does every match have its window?).  The reference's own narrowing this stands for:
DFAClassBuilder.java:365-376 (prefix indexOf), :420-426 (first-byte mask), CompilationPolicy.java:44-57.  No GPU needed."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_dictionary_gets_a_filter_and_it_is_exact(oracle_lib):
    """C3-sparse (1000 keywords of 6..8 chars, the compressed automaton): stride 2, run-up 8 = max_len, ~2000 windows."""
    from needle_amd import workload as W
    from needle_amd.pattern import DFACompiler
    from test_compile_matches_txt import oracle_for
    import prefilter_sim as sim
    words = W.keywords(1000, min_len=6, max_len=8)
    rx = "|".join(words)
    p = DFACompiler.compile(rx, "t", 0)
    o, _ = oracle_for(rx, 0)
    for which in ("contained_in", "forwards"):
        i = p.prefilter_info(which)
        assert i["on"] == 1 and i["mode"] == 6 and i["stride"] == 2 and i["warm"] == 8 and i["min_len"] == 6, i
        assert 1500 <= i["n_windows"] <= 2000 and i["bitmap_bytes"] == 32768, i
        assert i["on2"] == 1 and 1500 <= i["n_windows2"] <= 2100 and i["bitmap2_bytes"] == 8192, i  # the second level: 5-byte windows
    rows = W.keyword_batch(np, words, 11, 72, 256)
    rows[::7, 256 - len(words[3]):] = [ord(c) for c in words[3]]        # a keyword that ends with the row
    rows[3::7, 256 - len(words[4]) + 1:] = [ord(c) for c in words[4]][:-1]  # ... and one the row's end cuts
    rows[5::7, :len(words[9])] = [ord(c) for c in words[9]]              # ... and one at the very start
    lens = (np.arange(len(rows)) * 37 % 257).astype(np.uint32)
    fi = p.prefilter_info("forwards", with_bitmap=True)
    ci = p.prefilter_info("contained_in", with_bitmap=True)
    n_match = 0
    for k, row in enumerate(rows):
        for text in (row, row[:lens[k]]):
            want = o.find_all(text)[:1]
            for phase in range(fi["stride"]):  # (whatever phase the sampled window ends have inside a row)
                got = sim.filtered(p, "find", text, info=fi, phase=phase)
                assert got == ((True,) + want[0] if want else (False, -1, -1)), (k, len(text), phase, got, want)
                assert sim.filtered(p, "contained_in", text, info=ci, phase=phase)[0] == bool(want)
            n_match += bool(want)
        if k % 8 == 0:  # every window a candidate: the restart K chars ahead alone
            want = o.find_all(row)[:1]
            assert sim.filtered(p, "find", row, all_windows=True, info=fi) == ((True,) + want[0] if want else (False, -1, -1))
    assert n_match > 25


@pytest.mark.parametrize("regex,why", [("[0-9]+", ""), ("abc", ""), ("a.*bcdefg", ""), ("(abcdef)+x", "")])
def test_small_or_unbounded_patterns_keep_the_ordinary_kernel(regex, why):
    """Default level: the filter is for automata in the compressed form only; everything else reports on = 0 and a reason."""
    from needle_amd.pattern import DFACompiler
    p = DFACompiler.compile(regex, "t", 0)
    for which in ("contained_in", "forwards"):
        i = p.prefilter_info(which)
        assert i["on"] == 0 and i["why"], i


CHILD = r'''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from needle_amd.pattern import DFACompiler
from test_compile_matches_txt import oracle_for
import prefilter_sim as sim
from oracle import walker
walker.build()
rng = np.random.default_rng(5)
# (regex, filter expected for find, for containedIn)
CASES = [("Sherlock|Holmes|Watson|Irene|Adler|John|Baker", False, False),   # min_len 4 < 5: no stride fits
         ("Sherlock|Holmes|Watson|Moriarty|Mycroft", True, True),
         ("Sherlock|Holmes|Watson|Irene|Adler|Baker", True, True),           # min_len 5: stride 2 with the TWO-SIDED second level (round 6)
         ("abcdef|bcdefgh|cdefghij|xabcde", True, True),                     # keywords inside / overlapping one another
         ("[Ss]herlock|[Hh]olmes(es)?", True, True),
         ("(foo|foobar|bar|barbaz|baz)quux", True, True),
         ("hello[0-9][0-9]world|[a-c]{6}", True, True),
         ("abcdefgh", True, True),                                            # one length: start = end - 8 (no lengths program needed)
         ("http://[a-z]{3}\\.com", True, True),
         ("a[0-9]+bcdefg", None, None),                                       # unbounded: whatever the analysis says must be exact
         ("abcde.{0,3}fghij", None, None)]
n_on = 0
two_sided = set()
for rx, want_f, want_c in CASES:
    p = DFACompiler.compile(rx, "t", 0)
    o, _ = oracle_for(rx, 0)
    alpha = sorted(set(ord(c) for c in rx if c.isalnum() or c in ":/. ")) + [ord(c) for c in " xyz019_"] + [200]
    fi, ci = p.prefilter_info("forwards", with_bitmap=True), p.prefilter_info("contained_in", with_bitmap=True)
    if want_f is not None:
        assert bool(fi["on"]) == want_f, (rx, fi)
        assert bool(ci["on"]) == want_c, (rx, ci)
    if not (fi["on"] or ci["on"]):
        assert fi["why"] and ci["why"]
        continue
    n_on += 1
    if fi["on"] and fi["on2"] == 2:
        two_sided.add((fi["stride"], fi["min_len"]))
    # texts built from pieces of the regex's own literals, so that near misses, overlaps and matches at both row ends are common
    lits = [w for w in rx.replace("(", "|").replace(")", "|").replace("?", "|").replace("[", "|").replace("]", "|").split("|") if w.isalnum()]
    for trial in range(100):
        parts = []
        while sum(map(len, parts)) < 60:
            k = rng.integers(0, 4)
            if k == 0 and lits:
                w = lits[rng.integers(0, len(lits))]
                a = rng.integers(0, len(w)); b = rng.integers(a, len(w) + 1)
                parts.append([ord(c) for c in (w if rng.integers(0, 2) else w[a:b])])
            else:
                parts.append(list(rng.choice(alpha, size=rng.integers(1, 6))))
        text = np.array(sum(parts, []), dtype=np.uint8)[:int(rng.integers(0, 64))]
        want = o.find_all(text)[:1]
        exp = ((True,) + want[0]) if want else (False, -1, -1)
        ph = trial % 3
        if fi["on"]:
            assert sim.filtered(p, "find", text, info=fi, phase=ph) == exp, (rx, bytes(text), exp)
            assert sim.filtered(p, "find", text, all_windows=True, info=fi, phase=ph) == exp, (rx, bytes(text), exp)
        if ci["on"]:
            assert sim.filtered(p, "contained_in", text, info=ci, phase=ph)[0] == bool(want), (rx, bytes(text))
assert n_on >= 8, n_on
assert (2, 5) in two_sided and (4, 7) in two_sided, two_sided  # shortest match 5 -> stride 2, 7 -> stride 4: both with the two-sided second level
# find-all behind the filter: a candidate's run that CROSSES an earlier accept and lives on (the search automaton keeps the
# higher-priority longer alternative and drops the restart threads) says nothing about its window -- filed as unknown, re-run
# from the row's cursor (ADVICE r4: `international|inter|nation` on "internationa..": "nation" was lost)
rx = "international|inter|nation|qrstuvwxyzab"
p = DFACompiler.compile(rx, "t", 0)
o, _ = oracle_for(rx, 0)
fi = p.prefilter_info("forwards", with_bitmap=True)
assert fi["on"], fi
lost = 0
texts = [b"internationa xyz", b"internationwide", b"  internationalinternationx", b"xinternationinternational nation inter", b"internation", b"internatio nation"]
for trial in range(300):
    parts = []
    while sum(map(len, parts)) < 50:
        w = ["international", "inter", "nation", "internation", "internationa", "nationa", " ", "x", "qrstuvwxyzab", "tion"][rng.integers(0, 10)]
        parts.append(w)
    texts.append("".join(parts).encode()[:int(rng.integers(10, 64))])
for tx in texts:
    text = np.frombuffer(tx, dtype=np.uint8)
    want = o.find_all(text)
    for aw in (False, True):
        assert sim.filtered_find_all(p, text, info=fi, all_windows=aw) == want, (tx, aw, want)
    lost += sim.filtered_find_all(p, text, info=fi, with_crossed=False) != want
assert lost > 0  # (the round-4 logic does lose matches on these texts)
for rx2 in ("Sherlock|Holmes|Watson|Moriarty|Mycroft", "abcdef|bcdefgh|cdefghij|xabcde", "(foo|foobar|bar|barbaz|baz)quux"):
    p2 = DFACompiler.compile(rx2, "t", 0)
    o2, _ = oracle_for(rx2, 0)
    f2 = p2.prefilter_info("forwards", with_bitmap=True)
    lits = [w for w in rx2.replace("(", "|").replace(")", "|").split("|") if w.isalnum()]
    for trial in range(150):
        parts = []
        while sum(map(len, parts)) < 60:
            w = lits[rng.integers(0, len(lits))]
            parts.append(w if rng.integers(0, 3) else w[:int(rng.integers(1, len(w) + 1))] + "q"[:int(rng.integers(0, 2))])
        text = np.frombuffer("".join(parts).encode()[:int(rng.integers(0, 64))], dtype=np.uint8)
        for aw in (False, True):
            assert sim.filtered_find_all(p2, text, info=f2, all_windows=aw) == o2.find_all(text), (rx2, bytes(text), aw)
print("PREFILTER-SIM-OK", n_on)
'''


def test_filter_algorithm_on_everyday_patterns_vs_oracle():
    """NEEDLE_PREFILTER=2 builds filters for plain LDS-table automata too (read once per process: a child)."""
    env = dict(os.environ, NEEDLE_PREFILTER="2", NEEDLE_PAIR_MAX_BYTES="0")  # (no pair tables: plain uint8 tables, which level 2 covers)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert "PREFILTER-SIM-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_switch_off():
    code = ("import sys; sys.path.insert(0, '.')\nfrom needle_amd import workload as W\nfrom needle_amd.pattern import DFACompiler\n"
            "p = DFACompiler.compile('|'.join(W.keywords(1000, min_len=6, max_len=8)), 't', 0)\n"
            "i = p.prefilter_info('forwards'); assert i['on'] == 0 and 'NEEDLE_PREFILTER=0' in i['why'], i; print('OFF-OK')")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, NEEDLE_PREFILTER="0"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert "OFF-OK" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
