"""GPU parity for the BASELINE.json configs, patterns built by the product's own table generator (needle_compile):
C1 http://.+ matches() on 1k short strings, C2 [0-9]+ containedIn(), C3 1k-keyword union find(), C5 BMP class
regex over UTF-16, plus every matches.txt golden row through the single-haystack Matcher mirror.  Oracle =
oracle/needle_walk.c fed with the SAME tables (read back through the C ABI); bit-exact."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_compile_matches_txt import flag_sets, oracle_for
from test_gpu_parity import gpu_run, rows_from_strings


def compiled(pattern, flags=0):
    from needle_amd.pattern import DFACompiler
    o, _ = oracle_for(pattern, flags)
    return DFACompiler.compile(pattern, "t", flags), o


@pytest.mark.gpu
def test_c1_url_matches_1k_strings():
    from needle_amd import workload as W
    p, o = compiled("http://.+")
    hs = W.url_strings(1000)
    rows, lens = rows_from_strings(hs, np.uint8)
    m, c, f, fs, fe = gpu_run(p, rows, lens)
    want = np.array([o.matches(h) for h in hs])
    assert (m == want).all()
    assert 400 < want.sum() < 600  # 50 % well-formed; the newline / no-prefix variants must not match
    assert (c == np.array([o.contained_in(h) for h in hs])).all()
    for i, h in enumerate(hs):
        found, s, e = o.find(h)
        assert f[i] == found and (not found or (fs[i], fe[i]) == (s, e))
    # DFACompilerTest.java:524-540
    mm = p.matcher("http://www.google.com")
    assert mm.matches() and mm.containedIn() and mm.find() and (mm.start(), mm.end()) == (0, 21)
    mm = p.matcher("http://Γειά σου.com")
    assert mm.find() and (mm.start(), mm.end()) == (0, 19)


@pytest.mark.gpu
@pytest.mark.parametrize("ragged", [False, True])
def test_c2_digits_contained_in(ragged):
    import torch
    from needle_amd import workload as W
    from needle_amd.pattern import unpack_bitmap
    p, o = compiled("[0-9]+")
    n = 200_003
    rows = W.digits_batch(torch, 12345, n, 256, device="cuda")
    host = rows.cpu().numpy()
    assert (host == W.digits_batch(np, 12345, n, 256)).all()
    lens = None
    tl = None
    if ragged:
        lens = (np.arange(n, dtype=np.uint32) * 2654435761 % 257).astype(np.uint32)
        tl = torch.from_numpy(lens.astype(np.int32)).cuda()
    got = unpack_bitmap(p.contained_in_batch(rows, tl), n)
    want = o.batch_contained_in(host, lens, threads=4)
    assert (got == want).all()
    if not ragged:
        assert abs(want.mean() - 0.5) < 0.01
    fw, fs, fe = p.find_batch(rows, tl)
    of, ofs, ofe = o.batch_find(host, lens, threads=4)
    assert (unpack_bitmap(fw, n) == of).all() and (fs.cpu().numpy() == ofs).all() and (fe.cpu().numpy() == ofe).all()
    assert (unpack_bitmap(p.matches_batch(rows, tl), n) == o.batch_matches(host, lens, threads=4)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("ragged", [False, True])
def test_c3_keyword_union_find(ragged):
    import torch
    from needle_amd import workload as W
    from needle_amd.pattern import unpack_bitmap
    words = W.keywords(1000)
    p, o = compiled("|".join(words))
    assert p.info()["n_states"]["forwards"] > 1000  # LDS table stress: uint16 table mode
    n = 100_000
    rows = W.keyword_batch(torch, words, 777, n, 256, device="cuda")
    host = rows.cpu().numpy()
    lens = tl = None
    if ragged:
        lens = (np.arange(n, dtype=np.uint32) * 40503 % 257).astype(np.uint32)
        tl = torch.from_numpy(lens.astype(np.int32)).cuda()
    fw, fs, fe = p.find_batch(rows, tl)
    of, ofs, ofe = o.batch_find(host, lens, threads=4)
    assert (unpack_bitmap(fw, n) == of).all()
    assert (fs.cpu().numpy() == ofs).all() and (fe.cpu().numpy() == ofe).all()
    assert of.mean() > 0.25  # planted rows + chance hits
    assert (unpack_bitmap(p.contained_in_batch(rows, tl), n) == o.batch_contained_in(host, lens, threads=4)).all()
    assert (unpack_bitmap(p.matches_batch(rows, tl), n) == o.batch_matches(host, lens, threads=4)).all()


@pytest.mark.gpu
def test_c5_bmp_class_regex_utf16():
    import torch
    from needle_amd import workload as W
    from needle_amd.pattern import unpack_bitmap
    p, o = compiled(W.script_regex())
    n = 100_000
    rows = W.script_batch(torch, 99, n, 256, device="cuda")
    host = rows.cpu().numpy().view(np.uint16)
    assert (host == W.script_batch(np, 99, n, 256)).all()
    fw, fs, fe = p.find_batch(rows)
    of, ofs, ofe = o.batch_find(host, threads=4)
    assert (unpack_bitmap(fw, n) == of).all()
    assert (fs.cpu().numpy() == ofs).all() and (fe.cpu().numpy() == ofe).all()
    assert 0.25 < of.mean() < 0.6
    assert (unpack_bitmap(p.contained_in_batch(rows), n) == o.batch_contained_in(host, threads=4)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("ragged", [False, True])
def test_c5w_multi_class_bmp_regex_utf16(ragged):
    """C5's wide variant (needle_amd/workload.py SEQ_ALTS): 30 char classes, 33 states -- UTF-16 rows through the two-level page
    map (DFA.java:438-463 byte classes; DFAClassBuilder.java:269-305 the class lookup) into an LDS table.  Against the oracle on
    the same tables and, on a sample, against Python's `re` on the same regex (an implementation that shares nothing with either)."""
    import re
    import torch
    from needle_amd import workload as W
    from needle_amd.pattern import unpack_bitmap
    rx = W.scriptseq_regex()
    p, o = compiled(rx)
    inf = p.info()
    assert inf["stride"] >= 20 and inf["n_states"]["forwards"] > 5
    n = 100_000
    rows = W.scriptseq_batch(torch, 41, n, 256, device="cuda")
    host = rows.cpu().numpy().view(np.uint16)
    assert (host == W.scriptseq_batch(np, 41, n, 256)).all()
    lens = tl = None
    if ragged:
        lens = (np.arange(n, dtype=np.uint32) * 2654435761 % 257).astype(np.uint32)
        tl = torch.from_numpy(lens.astype(np.int32)).cuda()
    fw, fs, fe = p.find_batch(rows, tl)
    of, ofs, ofe = o.batch_find(host, lens, threads=4)
    fs, fe = fs.cpu().numpy(), fe.cpu().numpy()
    assert (unpack_bitmap(fw, n) == of).all()
    assert (fs == ofs).all() and (fe == ofe).all()
    assert (0.15 if ragged else 0.25) < of.mean() < 0.6
    assert (unpack_bitmap(p.contained_in_batch(rows, tl), n) == o.batch_contained_in(host, lens, threads=4)).all()
    assert (unpack_bitmap(p.matches_batch(rows, tl), n) == o.batch_matches(host, lens, threads=4)).all()
    cre = re.compile(rx)
    for i in range(0, n, 97):
        text = "".join(map(chr, host[i][:None if lens is None else lens[i]]))
        mm = cre.search(text)
        assert ((True, mm.start(), mm.end()) if mm else (False, -1, -1)) == (bool(of[i]), int(fs[i]), int(fe[i])), i
    # every match of every row (one-pass find-all) on a slice, against the oracle's repeated find()
    k = 20_000
    offs, as_, ae = p.find_all_batch(rows[:k], None if tl is None else tl[:k])
    offs, as_, ae = offs.cpu().numpy(), as_.cpu().numpy(), ae.cpu().numpy()
    for i in range(0, k, 53):
        want = o.find_all(host[i] if lens is None else host[i][:lens[i]])
        assert list(zip(as_[offs[i]:offs[i + 1]].tolist(), ae[offs[i]:offs[i + 1]].tolist())) == want, i


@pytest.mark.gpu
def test_c5w_with_the_compact_two_level_page_map():
    """The same C5w checks with NEEDLE_FLAT_MAP=0 (read once per process: a child): ptab -> page -> column, 128-byte tiles."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_configs.py", "-x", "-q", "-m", "gpu", "-k", "c5w_multi_class"],
                       env=dict(os.environ, NEEDLE_FLAT_MAP="0"), capture_output=True, text=True, timeout=1500, cwd=root)
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_matches_txt_rows_through_gpu_matcher():
    doc = json.load(open(os.path.join(GOLDEN, "matches.json")))
    from needle_amd.pattern import DFACompiler
    cache = {}
    n = 0
    for row in doc["rows"]:
        flags = flag_sets(row)[0]
        key = (row["pattern"], flags)
        if key not in cache:
            cache[key] = DFACompiler.compile(row["pattern"], "t", flags)
        m = cache[key].matcher(row["haystack"])
        assert m.find() == row["found"], row
        if row["found"]:
            assert (m.start(), m.end()) == (row["start"], row["end"]), row
        n += 1
    assert n > 180
    for case in doc["inline"]:
        p = DFACompiler.compile(case["pattern"], "t", case["flags"])
        m = p.matcher(case["haystack"])
        if "matches" in case:
            assert m.matches() == case["matches"]
        if "find" in case:
            rng = case.get("find_range")
            found = m.find(*rng) if rng else m.find()
            assert [found, m.start(), m.end()][:1] == case["find"][:1]
            if found:
                assert [m.start(), m.end()] == case["find"][1:]


@pytest.mark.gpu
def test_host_buffer_entry_points_match_device_entry_points():
    from needle_amd.pattern import unpack_bitmap
    p, o = compiled("[0-9]+")
    rng = np.random.default_rng(7)
    rows = rng.choice(np.frombuffer(b"abc 019xyz", dtype=np.uint8), size=(5000, 100))  # stride not a multiple of 16
    lens = rng.integers(0, 101, size=5000).astype(np.uint32)
    w = p.contained_in_batch(rows, lens)
    assert (unpack_bitmap(w, 5000) == o.batch_contained_in(rows, lens)).all()
    fw, fs, fe = p.find_batch(rows, lens)
    of, ofs, ofe = o.batch_find(rows, lens)
    assert (unpack_bitmap(fw, 5000) == of).all() and (fs == ofs).all() and (fe == ofe).all()


@pytest.mark.gpu
@pytest.mark.parametrize("hybrid,lds,mode", [("0", "4096", 3), ("1", "4096", 5), ("1", "20000", 5), ("1", "20000", 6), ("1", "12000", 6)])
def test_hbm_resident_table_modes_match_oracle(hybrid, lds, mode):
    """Automata too large for the LDS: forced through NEEDLE_MAX_PROG_LDS in a child process on a keyword-union
    pattern; same parity bar as the LDS-table mode.  Mode 3 = whole table walked out of HBM / L2 (NEEDLE_HYBRID=0);
    mode 5 = the rows of the first states in breadth-first order in LDS (a few dozen rows at 4 KiB -- most steps then take
    the cold path through the scalar cache -- or about half the automaton at 20 KB), the whole table in HBM; mode 6 = the
    compressed automaton (dense rows near the start state + exception records, chains included at 12 KB), the first
    choice when it fits -- the other modes are then reached with NEEDLE_SPARSE=0.  UTF-16 rows of the same text too."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler, unpack_bitmap
from test_compile_matches_txt import oracle_for
words = W.keywords(300)
rx = "|".join(words)
p = DFACompiler.compile(rx, "t", 0)
assert p.info()["kernel_mode"]["forwards"] == int(sys.argv[1]), p.info()
o, _ = oracle_for(rx, 0)
n = 20000
rows = W.keyword_batch(torch, words, 5, n, 256, device="cuda")
host = rows.cpu().numpy()
fw, fs, fe = p.find_batch(rows)
of, ofs, ofe = o.batch_find(host, threads=4)
assert (unpack_bitmap(fw, n) == of).all() and (fs.cpu().numpy() == ofs).all() and (fe.cpu().numpy() == ofe).all()
assert (unpack_bitmap(p.contained_in_batch(rows), n) == o.batch_contained_in(host, threads=4)).all()
assert (unpack_bitmap(p.matches_batch(rows), n) == o.batch_matches(host, threads=4)).all()
rows16 = rows.to(torch.int16)
fw, fs, fe = p.find_batch(rows16)
assert (unpack_bitmap(fw, n) == of).all() and (fs.cpu().numpy() == ofs).all() and (fe.cpu().numpy() == ofe).all()
print("GLOBAL-MODE-OK")
'''
    import os
    env = dict(os.environ, NEEDLE_MAX_PROG_LDS=lds, NEEDLE_HYBRID=hybrid, NEEDLE_SPARSE="1" if mode == 6 else "0")
    r = subprocess.run([sys.executable, "-c", code, str(mode)], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert "GLOBAL-MODE-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("name,regex", [("DigitPlus", "[0-9]+"), ("UnionOfManyNames", "Sherlock|Holmes|Watson|Irene|Adler|John|Baker"),
                                        ("RepeatingUnionOfShortStrings", "(ab|a|bcdef|g)+"), ("aDotc", "a.c")])
@pytest.mark.parametrize("ragged", [False, True])
def test_find_all_equals_repeated_reference_find(name, regex, ragged):
    """the nextStart cursor (DFAClassBuilder.java:616-659).  Per row: every non-overlapping match, GPU
    rounds of needle_find_next_dev vs the oracle's repeated find(); plus the second find() the reference's own
    compiled classes produced (tests/golden/snapshots 'find2')."""
    import torch
    from conftest import load_snapshot
    doc = load_snapshot(name)
    p, o = compiled(regex)
    hs = [v["h"] for v in doc["vectors"] if len(v["h"]) <= 64 and all(ord(c) < 256 for c in v["h"])]
    rng = np.random.default_rng(5)
    rows, lens = rows_from_strings(hs, np.uint8, stride=64)
    if not ragged:  # fill the padding with text so that full-length rows carry several matches
        alpha = np.frombuffer((regex + " xyz019").encode("latin-1", "ignore"), dtype=np.uint8)
        alpha = alpha[(alpha >= 48)]
        for i, h in enumerate(hs):
            rows[i, len(h):] = rng.choice(alpha, size=64 - len(h))
        lens = None
    t = torch.from_numpy(rows).cuda()
    tl = None if lens is None else torch.from_numpy(lens.astype(np.int32)).cuda()
    offsets, st, en = p.find_all_batch(t, tl)
    offsets, st, en = offsets.cpu().numpy(), st.cpu().numpy(), en.cpu().numpy()
    total = 0
    for i in range(len(hs)):
        row = rows[i] if lens is None else rows[i, :lens[i]]
        want = o.find_all(row)
        got = list(zip(st[offsets[i]:offsets[i + 1]].tolist(), en[offsets[i]:offsets[i + 1]].tolist()))
        assert got == want, (i, hs[i], got, want)
        total += len(want)
    assert total > len(hs) // 4
    # the C ABI's own find-all (needle_find_all_dev: dense per-row slots): same matches; with too few slots it files the
    # first ones and says that there were more
    most = max(int(offsets[i + 1] - offsets[i]) for i in range(len(hs)))
    for slots in (most, max(1, most - 1)):
        counts, ds, de, more = p.find_all_dense(t, slots, tl)
        counts, ds, de = counts.cpu().numpy(), ds.cpu().numpy(), de.cpu().numpy()
        if slots == most:
            counts_full, ds_full, de_full = counts, ds, de
        assert more == (slots < most)
        for i in range(len(hs)):
            k = int(offsets[i + 1] - offsets[i])
            assert counts[i] == min(k, slots), i
            assert ds[i, :counts[i]].tolist() == st[offsets[i]:offsets[i] + counts[i]].tolist(), i
            assert de[i, :counts[i]].tolist() == en[offsets[i]:offsets[i] + counts[i]].tolist(), i
    hc, hs_, he, hmore = p.find_all_dense(rows, most, lens)  # numpy rows: needle_find_all_host
    assert not hmore and (hc == counts_full).all() and (hs_ == ds_full).all() and (he == de_full).all()
    if ragged:  # the reference's own second find()
        by_h = {h: i for i, h in enumerate(hs)}
        for v in doc["vectors"]:
            if v["h"] in by_h and "find2" in v:
                i = by_h[v["h"]]
                got = list(zip(st[offsets[i]:offsets[i + 1]].tolist(), en[offsets[i]:offsets[i + 1]].tolist()))
                assert got[0] == (v["find"][1], v["find"][2])
                if v["find2"][0]:
                    assert got[1] == (v["find2"][1], v["find2"][2]), (v, got)
                else:
                    assert len(got) == 1, (v, got)


@pytest.mark.gpu
def test_find_next_with_cursor_edge_cases():
    """Cursors < 0 (exhausted), == length, mid-row; root-accepting pattern (a*) keeps the reference's literal-0 quirk."""
    import torch
    p, o = compiled("[0-9]+")
    hs = ["ab12cd345", "12", "", "x9", "99999999"]
    rows, lens = rows_from_strings(hs, np.uint8, stride=16)
    t = torch.from_numpy(rows).cuda()
    tl = torch.from_numpy(lens.astype(np.int32)).cuda()
    for cur in ([0, 0, 0, 0, 0], [4, 2, 0, 2, 3], [-1, 1, -1, 1, 8], [9, 5, 1, 0, 7]):
        c = torch.tensor(cur, dtype=torch.int32, device="cuda")
        _, st, en = p.find_next_batch(t, c, tl)
        for i, h in enumerate(hs):
            if cur[i] < 0:
                want = (-1, -1)
            else:
                f, s, e = o.find(h, start=cur[i])
                want = (s, e) if f else (-1, -1)
            assert (int(st[i]), int(en[i])) == want, (h, cur[i], int(st[i]), int(en[i]), want)
    p2, o2 = compiled("a*")
    hs = ["baaa", "aab", "", "bbb"]
    rows, lens = rows_from_strings(hs, np.uint8, stride=16)
    t = torch.from_numpy(rows).cuda()
    tl = torch.from_numpy(lens.astype(np.int32)).cuda()
    for cur in ([0, 0, 0, 0], [1, 1, 0, 2], [4, 3, 0, 3]):
        c = torch.tensor(cur, dtype=torch.int32, device="cuda")
        _, st, en = p2.find_next_batch(t, c, tl)
        for i, h in enumerate(hs):
            f, s, e = o2.find(h, start=cur[i])
            assert f and (int(st[i]), int(en[i])) == (s, e), (h, cur[i], int(st[i]), int(en[i]), s, e)


NULLABLE_CASES = [("a*", ["aa", "baa", "aab", "", "b", "aaaa" * 8]), ("(ab)*", ["abab", "xabab", "ab", "aba", ""]),
                  ("(a|b)*c?", ["abc", "abab", "cab", "ccc"])]


@pytest.mark.parametrize("regex,hs", NULLABLE_CASES)
def test_oracle_find_all_terminates_on_nullable_patterns(regex, hs):
    """a nullable pattern whose non-empty match ends exactly at the row end made the repeated find() cycle
    (0,len),(len,0),...: the literal-0 lastMatch of DFAClassBuilder.java:356.  The enumeration ends where the cursor
    stops advancing and the wrapped pseudo-match (end < start) is dropped."""
    from needle_amd.pattern import DFACompiler
    from oracle.walker import Dfa, OraclePattern
    t = DFACompiler.compile(regex).tables()
    d = {k: Dfa(t["class_map"], t["stride"], v["table"], v["accepting"], v["max_char"]) for k, v in t["dfas"].items()}
    o = OraclePattern(d["matches"], d["contained_in"], d["forwards"], d["backwards"], t["fixed_len"], -1)
    for h in hs:
        got = o.find_all(h.encode("latin-1"), limit=1000)
        assert len(got) <= len(h) + 1, (regex, h, got[:8])
        assert all(s <= e for s, e in got), (regex, h, got)
        ends = [e for _, e in got]
        assert ends == sorted(ends)


@pytest.mark.gpu
@pytest.mark.parametrize("regex,hs", NULLABLE_CASES)
def test_find_all_terminates_on_nullable_patterns(regex, hs):
    """The same on the GPU: the CSR form (library rounds, then the Python rounds with max_rounds=None never reached),
    the dense slots of needle_find_all_dev / needle_find_all_host, all equal to the oracle's enumeration."""
    import torch
    p, o = compiled(regex)
    rows, lens = rows_from_strings(hs, np.uint8, stride=32)
    t = torch.from_numpy(rows).cuda()
    tl = torch.from_numpy(lens.astype(np.int32)).cuda()
    want = [o.find_all(rows[i, :lens[i]]) for i in range(len(hs))]
    offsets, st, en = p.find_all_batch(t, tl)
    offsets, st, en = offsets.cpu().numpy(), st.cpu().numpy(), en.cpu().numpy()
    for i in range(len(hs)):
        got = list(zip(st[offsets[i]:offsets[i + 1]].tolist(), en[offsets[i]:offsets[i + 1]].tolist()))
        assert got == want[i], (regex, hs[i], got, want[i])
    # the Python round loop (what find_all_batch falls into when a row has more than 64 matches), bounded
    offsets2, st2, en2 = p.find_all_batch(t, tl, max_rounds=40)
    assert offsets2.cpu().numpy().tolist() == offsets.tolist() and st2.cpu().numpy().tolist() == st.tolist()
    most = max(1, max(len(w) for w in want))
    for dense_rows, dense_lens in ((t, tl), (rows, lens)):
        counts, ds, de, more = p.find_all_dense(dense_rows, most, dense_lens)
        if not isinstance(counts, np.ndarray):
            counts, ds, de = counts.cpu().numpy(), ds.cpu().numpy(), de.cpu().numpy()
        assert not more
        for i in range(len(hs)):
            assert list(zip(ds[i, :counts[i]].tolist(), de[i, :counts[i]].tolist())) == want[i], (regex, hs[i])
