"""The find-all TRANSDUCER behind the lock-step find-all kernel (needle_amd/csrc/needle_lower.h lower_find_all_transducer,
needle_find_all_ls.hip): the reference's repeated find() -- "restart the search at the end of each match", DFAClassBuilder.java:
616-659 -- folded into the automaton.  Host-side only: the device program's LDS image is walked here byte for byte the way the
kernel walks it (char -> column offset by window clamp / column map / page map; entry = table[(entry >> 4) * row_bytes + column];
a non-zero low nibble = a match code; the row's end = one more transition on the PAD column) and compared, match by match, with
the CPU oracle's repeated find() (oracle/walker.py).  Patterns that do not allow a transducer must be refused."""
import numpy as np
import pytest

from test_compile_matches_txt import oracle_for


def walk_blob(ft, text, cw=1):
    """Every match of one row, as find_all_lockstep_kernel computes them from the program blob."""
    b = ft["blob"]
    u16 = lambda off: int(b[off]) | int(b[off + 1]) << 8
    ncols_e, pad_e = ft["n_cols"] * 2, ft["pad_col"] * 2
    if cw == 1:
        tbase = 512  # kLdsTable1 (needle_device.h)
    else:
        tbase = ft["off_table"] - ft["win_lo_e"]
    def col_of(c):
        if ft["window"]:
            return min(max(c * 2, ft["win_lo_e"]), ft["win_hi_e"])
        if cw == 1:
            return u16(2 * c)  # cmap16 at 0
        return int(b[512 + u16(2 * (c >> 8)) + (c & 255)])  # ptab16 at 0 (page * 256), pages at 512 (column * 2)
    def emit(code, pos, out):
        d = u16(ft["codes_off"] + 4 * code) | u16(ft["codes_off"] + 4 * code + 2) << 16  # (k + length) | k << 16
        out.append((pos - (d & 0xFFFF), pos - (d >> 16)))
    e, out = ft["start"] << 4, []
    for pos, c in enumerate(text):
        e = u16(tbase + (e >> 4) * ncols_e + col_of(int(c)))
        if e & 15:
            emit(e & 15, pos, out)
    e = u16(tbase + (e >> 4) * ncols_e + pad_e)
    if e & 15:
        emit(e & 15, len(text), out)
    assert e >> 4 == 0
    return out


KEYWORDS = ["bc|abc", "zzy|xyzzy", "abc|bc", r"(\D|(.|b)([a-cx-z]|1))c", r"A|(\w)?.([a-cx-z]|(B|[x-z0]))", r"[^0-9a-f]|c[a\d]",
            r"(y|((\x41|[a-zA-Z])[^0-9a-f]|(.|a)中|a))", "abc|bcd|cdefg|a|xyzzy|zzy", "Sherlock|Holmes|Watson|Irene|Adler|John|Baker",
            "(ab|a|bcdef|g)", "aab|ab|b", "ab?c|abc?d", "http://|https://|ftp|tp:", "[ab]c|a[bc]d|[abc]{4}", "(foo|foobar|bar|barbaz|baz)",
            "a.c|ab", "(foo|foobar|bar|barbaz|baz)x?", "abcdef|bcd|cdefgh|f", "ab|abcd|cdx|dxyz", "Sherlock", "abcdefgh|abcd"]


@pytest.mark.parametrize("regex", KEYWORDS)
def test_transducer_blob_equals_repeated_find(regex, oracle_lib):
    from needle_amd.pattern import DFACompiler
    p = DFACompiler.compile(regex, "t", 0)
    o, _ = oracle_for(regex, 0)
    alphabet = sorted(set(ord(c) for c in regex if ord(c) < 256 and (c.isalnum() or c in ":/. "))) + [ord(c) for c in " ~_\n019xyzAB"] + [200]
    rng = np.random.default_rng(7)
    for cw in (1, 2):
        ft = p.find_all_transducer(cw)
        assert ft is not None and ft["lds_bytes"] == ft["blob"].size, regex
        alpha = alphabet + ([0x4E2D, 0x416, 0xFFFF] if cw == 2 else [])
        for trial in range(300):
            n = int(rng.integers(0, 40))
            text = rng.choice(alpha, size=n).astype(np.uint8 if cw == 1 else np.uint16) if n else np.zeros(0, dtype=np.uint8)
            assert walk_blob(ft, text.tolist(), cw) == o.find_all(text), (regex, cw, text.tolist())


def test_transducer_of_the_bench_dictionary(oracle_lib):
    """C3's 1000-keyword union: every accepting state dies on every char, so the transducer is the lengths automaton with its dead
    states folded away: 1464 states, 3 codes (lengths 3, 4, 5; k = 0), window addressing -- 86 KB of LDS, room for 16 waves of tiles."""
    from needle_amd import workload as W
    from needle_amd.pattern import DFACompiler
    words = W.keywords(1000)
    p = DFACompiler.compile("|".join(words), "t", 0)
    ft = p.find_all_transducer(1)
    ml = p.match_length_automaton()
    assert ft is not None and ft["window"] == 1 and ft["n_states"] <= ml["n_states"] and ft["lds_bytes"] + 16 * 64 * 64 <= 160 * 1024
    codes = [int(ft["blob"][ft["codes_off"] + 4 * c]) | int(ft["blob"][ft["codes_off"] + 4 * c + 1]) << 8 for c in range(16)]
    assert sorted(c for c in codes if c) == [3, 4, 5] and all(codes[c] in (0, c >> 1) for c in range(16))  # (k = 0 throughout: code = length << 1 | 1)
    o, _ = oracle_for("|".join(words), 0)
    rows = W.keyword_batch(np, words, 5, 200, 256)
    lens = (np.arange(len(rows)) * 37 % 257)
    for k, r in enumerate(rows):
        assert walk_blob(ft, r.tolist()) == o.find_all(r)
        assert walk_blob(ft, r[:lens[k]].tolist()) == o.find_all(r[:lens[k]])


@pytest.mark.parametrize("regex", ["[0-9]+", "a.*b", "(ab)+", "a*", "x?", "[a-z][a-z]+", "international|inter|nation|qrstuvwxyzab"])
def test_patterns_without_a_transducer_are_refused(regex):
    """Unbounded / empty matches have no lengths automaton; `nation` accepts inside a live `international`: a shadow of a shadow."""
    from needle_amd.pattern import DFACompiler
    ft = DFACompiler.compile(regex, "t", 0).find_all_transducer(1)
    assert ft is None or ft["kind"] == 2  # (kind 2: round 6's RUN transducer -- `[0-9]+`, `(ab)+`, `[a-z][a-z]+`: tested below)


@pytest.mark.parametrize("seed", range(3))
def test_transducer_on_random_regexes(seed, oracle_lib):
    """Seeded random regexes x flag sets: wherever the lowering offers a transducer, walking its blob reports exactly the oracle's
    repeated find() on random haystacks, 8- and 16-bit."""
    import random
    from needle_amd.pattern import DFACompiler, PatternException
    from test_compile_vs_python_restatement import FLAG_SETS, random_regex
    rng = random.Random(9000 + seed)
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
        ft2, ft1 = p.find_all_transducer(2), p.find_all_transducer(1)
        if ft2 is None:
            continue
        offered += 1
        for trial in range(100):
            n = int(nrng.integers(0, 40))
            text = nrng.choice(alphabet, size=n).astype(np.uint16)
            walk2 = walk_runs_blob if ft2["kind"] == 2 else walk_blob  # (kind 2: the RUN transducer of patterns without bounded lengths)
            assert walk2(ft2, text.tolist(), 2) == [tuple(x) for x in o.find_all(text)], (regex, flags, text.tolist())
            if ft1 is not None:
                t8 = (text & 0xFF).astype(np.uint8)
                walk1 = walk_runs_blob if ft1["kind"] == 2 else walk_blob
                assert walk1(ft1, t8.tolist(), 1) == [tuple(x) for x in o.find_all(t8)], (regex, flags, t8.tolist())
    assert offered >= 5


def walk_runs_blob(ft, text, cw=1):
    """Every match of one row as find_all_lockstep_kernel<.., RUNS> computes them from the RUN transducer's blob: code bit 0 = a match ends
    in front of this char, bit 1 = this char may begin a run; the start of a match is the last such char in front of its end."""
    b = ft["blob"]
    u16 = lambda off: int(b[off]) | int(b[off + 1]) << 8
    ncols_e, pad_e = ft["n_cols"] * 2, ft["pad_col"] * 2
    tbase = 512 if cw == 1 else ft["off_table"] - ft["win_lo_e"]
    def col_of(c):
        if ft["window"]:
            return min(max(c * 2, ft["win_lo_e"]), ft["win_hi_e"])
        if cw == 1:
            return u16(2 * c)
        return int(b[512 + u16(2 * (c >> 8)) + (c & 255)])
    e, out, run_start = ft["start"] << 4, [], 0
    for pos, c in enumerate(text):
        e = u16(tbase + (e >> 4) * ncols_e + col_of(int(c)))
        if e & 1:
            out.append((run_start, pos))
        if e & 2:
            run_start = pos
    e = u16(tbase + (e >> 4) * ncols_e + pad_e)
    if e & 1:
        out.append((run_start, len(text)))
    assert e >> 4 == 0
    return out


RUNS = ["[0-9]+", "[a-c]{3}[a-c]*", "[0-9]+x", "a+b+", "[a-z]+[0-9]", "ab*", "[а-яa-c]{2}[а-яa-c0-9]*"]
# `x[0-9]+` on "xx5", `(ab)+` on "aab", `ab+` on "aab": the attempt of the run's first char dies and a later start lives on inside the run;
# `9*y`: the start state hides a live attempt (it loops on '9'); `a.*b`: a match stays pending over chars that do not extend it
NOT_RUNS = ["a[0-9]+bcdefg|ab", "ab+", "x[0-9]+", "(ab)+", "9*y", "[0-9]+(\\.[0-9]+)?", "[а-я]{2}[а-я0-9]*|[0-9]+", "http://.+", "[0-9]*", "a.*b", "international|inter|nation|qrstuvwxyzab", "(a|ab)(c|bcd)*x"]


@pytest.mark.parametrize("regex", RUNS)
def test_run_transducer_blob_equals_repeated_find(regex, oracle_lib):
    """Patterns without bounded match lengths whose matches are runs (BASELINE's C2 / C5 kind): the RUN transducer (needle_lower.h
    lower_find_all_runs) walked the kernel's way against the oracle's repeated find() (forward walk + indexBackwards per match)."""
    from needle_amd.pattern import DFACompiler
    p = DFACompiler.compile(regex, "t", 0)
    o, _ = oracle_for(regex, 0)
    assert p.match_length_automaton() is None or p.find_all_transducer(1)["kind"] == 2, regex
    alphabet = sorted(set(ord(c) for c in regex if ord(c) < 256 and c.isalnum())) + [ord(c) for c in " ~019abcxy"] + [200]
    rng = np.random.default_rng(11)
    n_matches = 0
    for cw in (1, 2):
        ft = p.find_all_transducer(cw)
        assert ft is not None and ft["kind"] == 2 and ft["lds_bytes"] == ft["blob"].size, regex
        alpha = alphabet + ([0x4E2D, 0x0431, 0x0436, 0xFFFF] if cw == 2 else [])
        for trial in range(400):
            n = int(rng.integers(0, 48))
            text = rng.choice(alpha, size=n).astype(np.uint8 if cw == 1 else np.uint16) if n else np.zeros(0, dtype=np.uint8)
            want = o.find_all(text)
            assert walk_runs_blob(ft, text.tolist(), cw) == [tuple(x) for x in want], (regex, cw, text.tolist())
            n_matches += len(want)
    assert n_matches > 200, (regex, n_matches)


@pytest.mark.parametrize("regex", NOT_RUNS)
def test_patterns_that_are_not_runs_are_refused(regex):
    """A later attempt alive inside a run (`ab+` on "aab"), a match pending over chars that do not extend it, nullable patterns: no RUN
    transducer -- the pattern keeps its lengths transducer if it has one, else the per-lane one-pass kernel."""
    from needle_amd.pattern import DFACompiler
    p = DFACompiler.compile(regex, "t", 0)
    ft = p.find_all_transducer(1)
    assert ft is None or ft["kind"] == 1, (regex, ft and ft["kind"])
