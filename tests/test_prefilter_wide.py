"""The WIDE n-gram filter's host analysis (needle_amd/csrc/needle_ngram_host.cpp with cmap16; needle_lower.cpp lower_filter_wide): UTF-16
rows of a pattern that lives on SEVERAL pages of the BMP -- Latin + Cyrillic + CJK keyword dictionaries.  The reference's class map covers
all 65 536 code units of any pattern (DFA.java:438-463, DFAClassBuilder.java:269-305) and its prefilters run on any String
(DFAClassBuilder.java:365-376).  As tests/test_prefilter.py: the filter ALGORITHM in plain Python (tests/prefilter_sim.py: windows of four
16-bit code units, hash u = dot2(x0, m1 | m2 << 16) + dot2(x1, m1b | m2b << 16), second level + unit(q - 5) * m3) on the reference-layout
tables against the CPU oracle -- with the real bitmaps (does every match have its window?) and with every window a candidate (is the
restart K chars ahead exact?).  No GPU needed."""
import numpy as np


def test_mixed_script_dictionary_gets_a_wide_filter_and_it_is_exact(oracle_lib):
    from needle_amd import workload as W
    from needle_amd.pattern import DFACompiler
    from test_compile_matches_txt import oracle_for
    import prefilter_sim as sim
    words = W.keywords_mixed(100)
    rx = "|".join(words)
    p = DFACompiler.compile(rx, "t", 0)
    o, _ = oracle_for(rx, 0)
    assert p.utf16_route() is None  # three scripts: no single page of the BMP -- the one-page route does not apply
    infos = {}
    for which in ("contained_in", "forwards"):
        i = p.prefilter_info(which, with_bitmap=True, wide=True)
        assert i["on"] == 1 and i["wide"] == 1 and i["mode"] == 3 and i["stride"] == 2 and i["warm"] == 8 and i["min_len"] == 6, i
        assert 500 <= i["n_windows"] <= 600 and i["on2"] == 1 and i["m1b"] and i["m2b"], i
        infos[which] = i
        assert p.prefilter_info(which)["wide"] == 0  # (the byte programs' filter is another question)
    rows = W.mixed_keyword_batch(np, words, 5, 96, 128)
    for k, w in ((3, words[3]), (4, words[4]), (9, words[2])):
        rows[k::11, 128 - len(w):] = [ord(c) for c in w]          # a keyword that ends with the row
    rows[6::11, 128 - len(words[5]) + 1:] = [ord(c) for c in words[5]][:-1]  # one the row's end cuts
    rows[8::11, :len(words[7])] = [ord(c) for c in words[7]]      # one at the very start
    rows[10::11, 40:40 + len(words[1])] = [ord(c) for c in words[1]]
    rows[10::11, 40 + len(words[1]):40 + len(words[1]) + len(words[8])] = [ord(c) for c in words[8]]  # two adjacent matches
    lens = (np.arange(len(rows)) * 37 % 129).astype(np.uint32)
    n_match = 0
    for k, row in enumerate(rows):
        for text in (row, row[:lens[k]]):
            want_all = o.find_all(text)
            want = want_all[:1]
            exp = ((True,) + want[0]) if want else (False, -1, -1)
            assert sim.filtered(p, "find", text, info=infos["forwards"]) == exp, (k, len(text), exp)
            assert sim.filtered(p, "contained_in", text, info=infos["contained_in"])[0] == bool(want)
            assert sim.filtered_find_all(p, text, info=infos["forwards"]) == [tuple(x) for x in want_all], (k, len(text), want_all)
            n_match += len(want_all)
        if k % 6 == 0:
            want = o.find_all(row)[:1]
            assert sim.filtered(p, "find", row, all_windows=True, info=infos["forwards"]) == (((True,) + want[0]) if want else (False, -1, -1))
    assert n_match > 60, n_match


def test_wide_filter_reasons():
    """Patterns without a usable wide filter say why; a one-page dictionary gets a wide filter too (it is simply not the route taken)."""
    from needle_amd.pattern import DFACompiler
    p = DFACompiler.compile("[а-я]+|[a-z]+一", "t", 0)
    i = p.prefilter_info("contained_in", wide=True)
    assert i["on"] == 0 and i["why"], i
    p = DFACompiler.compile("привет|hello世界|世界世界世", "t", 0)
    i = p.prefilter_info("contained_in", wide=True)
    assert i["on"] == 1 and i["wide"] == 1 and i["min_len"] == 5 and i["stride"] == 2, i
