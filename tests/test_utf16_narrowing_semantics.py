"""What lets UTF-16 rows run behind the BYTE program's filter (needle_api.cpp utf16_route, needle_ngram.h narrow16).  Round 4's rule: for a
pattern whose chars all lie below 0xFF -- the anchored automaton's maxChar < 0xFF -- a char above 0xFE is "beyond maxChar"
(DFAClassBuilder.java:440, :565: `c > maxChar`), and so is byte 0xFF: the reference's answers on UTF-16 rows are its answers on the rows
narrowed char by char to min(c, 0xFF) (first test).  Round 5's general rule -- the one the library ships: a pattern that lives on ONE page P of
the BMP (every char outside it is of the pattern's "other" class, and so is the page's char P << 8 | sub) gives, on any UTF-16 row, the
answers it gives on the row with every char outside the page replaced by P << 8 | sub -- which is what the page's byte program sees once the
kernel has narrowed the text (test_route_rule_on_the_oracle: patterns with sub != 0xFF, pages other than 0, negated classes).  Checked here on the CPU oracle alone (no device): matches / containedIn / find / repeated find on random dictionaries and
a few regexes, with chars above 0xFF planted next to, inside and in place of keyword chars."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_compile_matches_txt import oracle_for  # noqa: E402
from needle_amd import workload as W  # noqa: E402
from needle_amd.pattern import DFACompiler  # noqa: E402

CASES = [
    "|".join(W.keywords(300, min_len=6, max_len=8)),
    "|".join(W.keywords(120, min_len=3, max_len=5)),
    "Sherlock|Holmes|Watson|Moriarty",
    "[0-9]+[A-Z][a-z]{2}é",          # Latin-1 char in the pattern (0xE9 < 0xFF)
    "(foo|bar)[a-z]*baz",
    "[^a]bc",                             # a negated class reaches 0xFFFF: maxChar >= 0xFF, the route must stay off
    "abcÿd",                         # 0xFF itself in the pattern: off
]


@pytest.mark.parametrize("rx", CASES)
def test_oracle_on_utf16_rows_equals_oracle_on_narrowed_rows(rx):
    p = DFACompiler.compile(rx, "t", 0)
    o, _ = oracle_for(rx, 0)
    max_char = p.info()["max_char"]["matches"]
    rng = np.random.default_rng(len(rx))
    n, width = 600, 96
    alpha = np.array([ord(c) for c in "abcdefghijklmnopqrstuvwxyz 0123456789ABCZé"], dtype=np.uint16)
    rows = rng.choice(alpha, (n, width)).astype(np.uint16)
    words = [w for w in rx.split("|") if w.isalnum()] or ["foobaz", "123Abcé", "Sherlock", "xbc"]
    for r in range(0, n, 2):
        w = np.array([ord(c) for c in words[r % len(words)]], dtype=np.uint16)
        at = int(rng.integers(0, width - len(w) + 1))
        rows[r, at:at + len(w)] = w
        if r % 6 == 0: rows[r, at + int(rng.integers(0, len(w)))] |= 0x0100   # a keyword char's low byte under a high byte
    m = rng.random(rows.shape) < 0.03
    rows[m] = rng.integers(0x0100, 0xFFFF, size=int(m.sum()), dtype=np.uint16)
    rows[rng.random(rows.shape) < 0.01] = 0x00FF
    narrowed = np.minimum(rows, 0xFF).astype(np.uint8)
    same = True
    of, ofs, ofe = o.batch_find(rows)
    nf, nfs, nfe = o.batch_find(narrowed)
    same &= bool((of == nf).all() and (ofs == nfs).all() and (ofe == nfe).all())
    same &= bool((o.batch_contained_in(rows) == o.batch_contained_in(narrowed)).all())
    same &= all(o.find_all(rows[i]) == o.find_all(narrowed[i]) for i in range(0, n, 7))
    if max_char < 0xFF:
        assert of.sum() > 0 or "baz" in rx
        assert same, (rx[:30], max_char)
    # (patterns reaching 0xFF and beyond may or may not agree: the route is off for them -- utf16_filter_ok; nothing to assert but that
    # the criterion is what the library reports)
    assert (max_char < 0xFF) == (rx not in ("[^a]bc", "abcÿd")), (rx, max_char)


def test_route_reports_the_patterns_page():
    """needle_pattern_utf16_route (host-side, from the tables alone): page 0 / sub 0xFF for ASCII dictionaries, page 4 for Cyrillic ones with a
    substitute of the "other" class on that page, nothing for a pattern on two pages or one that leaves its page no "other" char."""
    def cyr(w): return "".join(chr(0x0430 + ord(c) - 97) for c in w)
    words = W.keywords(200, min_len=6, max_len=8)
    assert DFACompiler.compile("|".join(words), "t", 0).utf16_route() == (0, 0xFF)
    pc = DFACompiler.compile("|".join(cyr(w) for w in words), "t", 0)
    page, sub = pc.utf16_route()
    assert page == 4 and not (0x30 <= sub <= 0x49)   # (0x0430 .. 0x0449 are the pattern's own letters a .. z)
    assert DFACompiler.compile("|".join(words[:50] + [cyr(w) for w in words[50:100]]), "t", 0).utf16_route() is None
    assert DFACompiler.compile("[Ѐ-ӿ]{4}[Ѐ-ӿ]*x?", "t", 0).utf16_route() is None  # two pages (x), and page 4 has no other char
    r = DFACompiler.compile("abcdefÿgh|bcdefgh", "t", 0).utf16_route()
    assert r is not None and r[0] == 0 and r[1] != 0xFF


ROUTED = [
    ("|".join("".join(chr(0x0430 + ord(c) - 97) for c in w) for w in W.keywords(150, min_len=5, max_len=8)), 4),   # Cyrillic dictionary: page 4
    ("abcdefÿgh|bcdefgh|[x-z]{5}", 0),                                                                         # 0xFF in the pattern: sub != 0xFF
    ("[α-ω]{3}[α-ω]*|λόγος", 3),                                                                                      # Greek: page 3
    ("(foo|bar)[a-z]*baz|[0-9]{4}", 0),
]


@pytest.mark.parametrize("rx,page", ROUTED)
def test_route_rule_on_the_oracle(rx, page):
    """For the route (page, sub) the library reports: oracle(rows) == oracle(rows with every char outside the page replaced by page << 8 | sub)
    for find / containedIn / repeated find -- chars of other pages planted next to, inside and in place of the pattern's chars, among them
    chars with the LOW BYTE of a pattern char under another high byte."""
    p = DFACompiler.compile(rx, "t", 0)
    o, _ = oracle_for(rx, 0)
    route = p.utf16_route()
    assert route is not None and route[0] == page, (rx[:30], route)
    sub = route[1]
    rng = np.random.default_rng(len(rx) + page)
    n, width = 500, 80
    own = np.array(sorted(set(ord(c) for c in rx if c.isalnum())), dtype=np.uint16)
    alpha = np.concatenate([own, np.array([32, 32, 48, 57, 0x41, 0x0416, 0x03A9, 0x4E2D], dtype=np.uint16)])
    rows = rng.choice(alpha, (n, width)).astype(np.uint16)
    words = [w for w in rx.split("|") if w.isalnum()] or ["abcdefgh"]
    for r in range(0, n, 2):
        w = np.array([ord(c) for c in words[r % len(words)]], dtype=np.uint16)[:width]
        at = int(rng.integers(0, width - len(w) + 1))
        rows[r, at:at + len(w)] = w
        if r % 6 == 0:
            rows[r, at + int(rng.integers(0, len(w)))] ^= 0x0100   # a pattern char's low byte on the neighbouring page
    m = rng.random(rows.shape) < 0.03
    rows[m] = rng.integers(0, 0xFFFF, size=int(m.sum()), dtype=np.uint16)
    replaced = np.where((rows >> 8) == page, rows, np.uint16(page << 8 | sub)).astype(np.uint16)
    of, ofs, ofe = o.batch_find(rows)
    nf, nfs, nfe = o.batch_find(replaced)
    assert of.sum() > 20, rx[:30]
    assert (of == nf).all() and (ofs == nfs).all() and (ofe == nfe).all(), rx[:30]
    assert (o.batch_contained_in(rows) == o.batch_contained_in(replaced)).all(), rx[:30]
    assert all(o.find_all(rows[i]) == o.find_all(replaced[i]) for i in range(0, n, 5)), rx[:30]
