"""SURVEY.md s8f-1 at batch scale: needle_find_all_dev's ONE-PASS kernel (needle_find_all.hip) -- every non-overlapping
match of every row, each lane restarting its search where its last match ended -- against the oracle's repeated find()
(oracle/walker.py find_all: Matcher.find() with the nextStart cursor, DFAClassBuilder.java:616-659), and against the
round-per-match form built on the scan kernel's per-row cursors.  Bit-exact: counts, starts, ends, the `more` flag."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DICTIONARY = r'''
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler
from test_compile_matches_txt import oracle_for
words = W.keywords(300)
rx = "|".join(words)
p = DFACompiler.compile(rx, "t", 0)
want_mode = int(sys.argv[1])
if want_mode >= 0:
    assert p.info()["kernel_mode"]["forwards"] == want_mode, p.info()
o, _ = oracle_for(rx, 0)
n = 3000
rows = W.keyword_batch(torch, words, 11, n, 256, device="cuda")
host = rows.cpu().numpy()
want = [o.find_all(host[i]) for i in range(n)]
most = max(len(w) for w in want)
assert most >= 6 and sum(len(w) for w in want) > n
for slots in (most, 3):
    counts, st, en, more = p.find_all_dense(rows, slots)
    counts, st, en = counts.cpu().numpy(), st.cpu().numpy(), en.cpu().numpy()
    assert more == (slots < most)
    for i in range(n):
        k = min(len(want[i]), slots)
        assert counts[i] == k, (i, counts[i], want[i])
        assert list(zip(st[i, :k].tolist(), en[i, :k].tolist())) == want[i][:k], (i, want[i])
        assert (st[i, k:] == -1).all() and (en[i, k:] == -1).all()  # slots beyond the count are untouched
# compact form: count pass, prefix sum, fill pass (needle_count_matches_dev / needle_find_all_csr_dev)
cnt = p.count_matches_batch(rows).cpu().numpy()
assert (cnt == [len(w) for w in want]).all()
offsets, s1, e1 = p.find_all_csr(rows)
offsets, s1, e1 = offsets.cpu().numpy(), s1.cpu().numpy(), e1.cpu().numpy()
assert offsets[-1] == sum(len(w) for w in want) == len(s1) == len(e1)
for i in range(n):
    assert list(zip(s1[offsets[i]:offsets[i + 1]].tolist(), e1[offsets[i]:offsets[i + 1]].tolist())) == want[i], i
# the round-per-match loop over the scan kernel's per-row cursors (find_next): an independent GPU path
offsets, s2, e2 = p.find_all_batch(rows, max_rounds=most + 2)
offsets, s2, e2 = offsets.cpu().numpy(), s2.cpu().numpy(), e2.cpu().numpy()
for i in range(0, n, 7):
    assert list(zip(s2[offsets[i]:offsets[i + 1]].tolist(), e2[offsets[i]:offsets[i + 1]].tolist())) == want[i]
# UTF-16 rows of the same text
rows16 = rows.to(torch.int16)
counts16, st16, en16, more16 = p.find_all_dense(rows16, most)
assert not more16 and (counts16.cpu().numpy() == [len(w) for w in want]).all()
st16, en16 = st16.cpu().numpy(), en16.cpu().numpy()
for i in range(0, n, 5):
    assert list(zip(st16[i, :len(want[i])].tolist(), en16[i, :len(want[i])].tolist())) == want[i]
print("FIND-ALL-OK")
'''


@pytest.mark.gpu
@pytest.mark.parametrize("env,mode", [({}, 2), ({"NEEDLE_WINDOW": "0"}, 2),
                                      ({"NEEDLE_FIND_ALL_LOCKSTEP": "0"}, 2), ({"NEEDLE_FIND_ALL_LOCKSTEP": "0", "NEEDLE_FIND_ALL_WINDOW": "0"}, 2), ({"NEEDLE_FIND_ALL_LENGTHS": "0"}, 2), ({"NEEDLE_FIND_ALL_LENGTHS": "0", "NEEDLE_FIND_ALL_DEFER": "0"}, 2),
                                      ({"NEEDLE_FIND_ALL_ROUNDS": "1"}, 2),
                                      ({"NEEDLE_MAX_PROG_LDS": "4096", "NEEDLE_HYBRID": "0"}, 3),
                                      ({"NEEDLE_MAX_PROG_LDS": "4096"}, 5), ({"NEEDLE_MAX_PROG_LDS": "20000", "NEEDLE_SPARSE": "0"}, 5),
                                      ({"NEEDLE_MAX_PROG_LDS": "20000", "NEEDLE_FIND_ALL_LENGTHS": "2"}, 6),
                                      ({"NEEDLE_MAX_PROG_LDS": "12000", "NEEDLE_WINDOW": "0", "NEEDLE_FIND_ALL_LENGTHS": "2"}, 6),
                                      ({"NEEDLE_MAX_PROG_LDS": "20000", "NEEDLE_FIND_ALL_ROUNDS": "1"}, 6),
                                      ({"NEEDLE_MAX_PROG_LDS": "12000", "NEEDLE_FIND_ALL_ROUNDS": "1", "NEEDLE_WINDOW": "0"}, 6)],
                         ids=["lock-step", "lock-step-column-maps", "one-pass-lengths", "one-pass-lengths-column-maps", "one-pass-backward-walks", "starts-at-once", "rounds", "hbm-table", "hot-rows-4k", "hot-rows-20k",
                              "one-pass-compressed-lengths", "one-pass-compressed-lengths-cmap",
                              "rounds-compressed-automaton", "rounds-compressed-automaton-cmap"])
def test_keyword_dictionary_every_match(env, mode):
    """A 300-keyword union (779 states, uint16 table) over 3000 rows of 256 chars: one to two matches per row, up to 8.
    Children: the lock-step kernel on the find-all transducer (the default for such a dictionary: needle_find_all_ls.hip; window
    addressing or column maps), the per-lane one-pass kernel with start = end - the length its end state remembers (needle_lower.h,
    MatchLengths), with the starts found by indexBackwards group by group or match by match, the round-per-match form,
    and the automaton forced out of the LDS (whole table in HBM; hot rows in LDS + HBM table; the compressed automaton -- in the
    one-pass kernel as the compressed LENGTHS program, window addressing or column maps, and in the round-per-match form, whose
    guarded kernels then carry the per-row cursors)."""
    r = subprocess.run([sys.executable, "-c", DICTIONARY, str(mode)], env=dict(os.environ, **env), capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert "FIND-ALL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def _oracle(regex, flags=0):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_compile_matches_txt import oracle_for
    from needle_amd.pattern import DFACompiler
    return DFACompiler.compile(regex, "t", flags), oracle_for(regex, flags)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("regex", ["[0-9]+", "(ab|a|bcdef|g)+", "a.c", "ab|a|bcdef|g", "[a-c]*", "abc|abcd1|d1"])
@pytest.mark.parametrize("stride", [16, 48, 64, 80, 192, 320, 1040])
def test_every_match_across_tile_boundaries(regex, stride):
    """Rows whose stride is not a multiple of the LDS tile (64 / 128 bytes), ragged lengths (empty rows included), text
    dense in matches: restarts, matches and backward walks straddle pieces, tiles and 128-byte lines.  `[a-c]*` matches
    the empty string (starts found match by match; an empty match ends its row); `abc|abcd1|d1` dies two chars after
    a match's end, so a restart steps back."""
    import torch
    p, o = _oracle(regex)
    rng = np.random.default_rng(stride)
    n = 700
    rows = rng.choice(np.frombuffer(b"abcdefg019 abca", dtype=np.uint8), size=(n, stride))
    lens = rng.integers(0, stride + 1, size=n).astype(np.uint32)
    lens[:50] = stride
    for ragged in (True, False):
        ln = lens if ragged else None
        want = [o.find_all(rows[i, :lens[i]] if ragged else rows[i]) for i in range(n)]
        most = max(1, max(len(w) for w in want))
        t = torch.from_numpy(rows).cuda()
        tl = torch.from_numpy(lens.astype(np.int32)).cuda() if ragged else None
        counts, st, en, more = p.find_all_dense(t, most, tl)
        counts, st, en = counts.cpu().numpy(), st.cpu().numpy(), en.cpu().numpy()
        assert not more
        for i in range(n):
            k = len(want[i])
            assert counts[i] == k, (i, counts[i], want[i][:4])
            assert list(zip(st[i, :k].tolist(), en[i, :k].tolist())) == want[i], (i, want[i][:4])
        if most > 1:
            c2, _, _, more2 = p.find_all_dense(t, most - 1, tl)
            assert more2 and (c2.cpu().numpy() == np.minimum(counts, most - 1)).all()
        # the compact form on host buffers (needle_find_all_csr_host; 700 rows of up to ~30 matches: the guessed capacity of
        # 1400 entries is too small for the dense strides, so the second call is exercised too)
        ho, hs2, he2 = p.find_all_csr(rows, ln)
        assert ho[-1] == sum(len(w) for w in want) == len(hs2)
        for i in range(0, n, 3):
            assert list(zip(hs2[ho[i]:ho[i + 1]].tolist(), he2[ho[i]:ho[i + 1]].tolist())) == want[i], i
        # compact form (count, prefix sum, fill) and the CSR wrapper built on it
        assert (p.count_matches_batch(t, tl).cpu().numpy() == [len(w) for w in want]).all()
        offsets, s1, e1 = p.find_all_batch(t, tl)
        offsets, s1, e1 = offsets.cpu().numpy(), s1.cpu().numpy(), e1.cpu().numpy()
        assert offsets[-1] == sum(len(w) for w in want)
        for i in range(n):
            assert list(zip(s1[offsets[i]:offsets[i + 1]].tolist(), e1[offsets[i]:offsets[i + 1]].tolist())) == want[i], i


@pytest.mark.gpu
def test_no_slots_and_no_matches():
    """max_per_row = 0 files nothing and only says whether any row had a match; a batch without matches has an empty
    compact form."""
    import torch
    p, o = _oracle("[0-9]+")
    rows = torch.from_numpy(np.frombuffer(b"no digits in here, none at all!!" * 200, dtype=np.uint8).reshape(200, 32).copy()).cuda()
    counts, st, en, more = p.find_all_dense(rows, 0)
    assert not more and int(counts.sum()) == 0
    offsets, s1, e1 = p.find_all_batch(rows)
    assert int(offsets[-1]) == 0 and s1.numel() == 0 and e1.numel() == 0
    assert int(p.count_matches_batch(rows).sum()) == 0
    rows[17, 5] = ord("7")
    counts, st, en, more = p.find_all_dense(rows, 0)
    assert more and int(counts.sum()) == 0
    cnt = p.count_matches_batch(rows).cpu().numpy()
    assert cnt[17] == 1 and cnt.sum() == 1
    offsets, s1, e1 = p.find_all_batch(rows)
    assert offsets.cpu().numpy()[17:19].tolist() == [0, 1] and s1.tolist() == [5] and e1.tolist() == [6]


@pytest.mark.gpu
def test_every_match_utf16_script_runs():
    """The C5 regex (runs of Greek / Cyrillic / Hebrew / CJK code units) over UTF-16 rows: packed functions behind the
    two-level page map, one to a few runs per row."""
    import torch
    from needle_amd import workload as W
    p, o = _oracle(W.script_regex())
    n = 1500
    rows = W.script_batch(torch, 3, n, 256, device="cuda")
    host = rows.cpu().numpy().view(np.uint16)
    want = [o.find_all(host[i]) for i in range(n)]
    most = max(len(w) for w in want)
    assert most >= 1
    counts, st, en, more = p.find_all_dense(rows, most + 1)
    counts, st, en = counts.cpu().numpy(), st.cpu().numpy(), en.cpu().numpy()
    assert not more
    for i in range(n):
        k = len(want[i])
        assert counts[i] == k and list(zip(st[i, :k].tolist(), en[i, :k].tolist())) == want[i], (i, want[i])


@pytest.mark.gpu
def test_every_match_in_rows_beyond_16_bit_indices():
    """Rows of 70 000 chars (indices beyond 16 bits, hundreds of tiles per row); and a row with more matches than slots
    says so."""
    import torch
    p, o = _oracle("[0-9]+x")
    rng = np.random.default_rng(9)
    L = 70000
    rows = rng.choice(np.frombuffer(b"abcdefghijklmnopqrstuvwxyz      01x", dtype=np.uint8), size=(6, L))
    rows[:, 66000:66004] = np.frombuffer(b"123x", dtype=np.uint8)
    want = [o.find_all(rows[i]) for i in range(6)]
    most = max(len(w) for w in want)
    assert most > 20 and any(e > 65536 for w in want for _, e in w)
    t = torch.from_numpy(rows).cuda()
    counts, st, en, more = p.find_all_dense(t, most)
    counts, st, en = counts.cpu().numpy(), st.cpu().numpy(), en.cpu().numpy()
    assert not more
    for i in range(6):
        k = len(want[i])
        assert counts[i] == k and list(zip(st[i, :k].tolist(), en[i, :k].tolist())) == want[i]
    c2, s2, e2, more2 = p.find_all_dense(t, 5)
    assert more2 and (c2.cpu().numpy() == 5).all() and (s2.cpu().numpy() == st[:, :5]).all() and (e2.cpu().numpy() == en[:, :5]).all()


DENSE_HOST = r'''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from needle_amd.pattern import DFACompiler
from test_compile_matches_txt import oracle_for
# a one-char pattern over rows of that char class: ~stride matches per row, i.e. results several times the row bytes
p = DFACompiler.compile("[a-f]", "t", 0)
o, _ = oracle_for("[a-f]", 0)
rng = np.random.default_rng(5)
n, stride = 3000, 96
rows = rng.choice(np.frombuffer(b"abcdefgh", dtype=np.uint8), size=(n, stride))
offs, st, en = p.find_all_csr(rows)               # needle_find_all_csr_host: the fill pass in sub-ranges of rows
want = [o.find_all(rows[i]) for i in range(0, n, 37)]
assert offs[-1] == len(st) == len(en) and offs[-1] > 4 * n * 8  # > 8 KiB budget many times over
for k, i in enumerate(range(0, n, 37)):
    assert list(zip(st[offs[i]:offs[i + 1]].tolist(), en[offs[i]:offs[i + 1]].tolist())) == want[k], i
assert (en - st == 1).all() and (np.diff(offs) == (rows < ord("g")).sum(1)).all()
print("DENSE-HOST-OK")
'''


@pytest.mark.gpu
def test_csr_host_dense_matches_bounded_results():
    """needle_find_all_csr_host on a dense-match batch with the device-resident result budget shrunk to 8 KiB: the fill pass
    runs over many sub-ranges of the chunk's rows and the pieces line up (results were unbounded)."""
    env = dict(os.environ, NEEDLE_HOST_RESULT_BYTES="8192")
    r = subprocess.run([sys.executable, "-c", DENSE_HOST], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert "DENSE-HOST-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("regex", ["Sherlock|Holmes|Watson|ab|abc|bc", "[0-9]+", "(ab|a|bcdef|g)+", "a*", "a.c"])
@pytest.mark.parametrize("cw", [1, 2])
def test_packed16_form_equals_the_two_array_form(regex, cw):
    """needle_find_all_packed16_dev (one dword per match: start | end << 16) files what needle_find_all_dev files -- the
    lengths-automaton walk, the deferred starts phase (the end travels in the high half until the backward walk adds the
    start) and the immediate form of nullable patterns; full and ragged rows, too few slots included."""
    import torch
    from needle_amd.pattern import DFACompiler
    p = DFACompiler.compile(regex, "t", 0)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    lut = torch.tensor([ord(c) for c in "abcdefg019 SherlockHmsWtn"], dtype=torch.uint8, device="cuda")
    n = 20011
    rows = lut[torch.randint(0, len(lut), (n, 112), device="cuda", generator=g)]
    if cw == 2:
        rows = rows.to(torch.int16)
    lens = torch.randint(0, 113, (n,), device="cuda", generator=g).to(torch.int32)
    for l in (None, lens):
        for slots in (40, 3):
            c0, s0, e0, m0 = p.find_all_dense(rows, slots, l)
            c1, se, m1 = p.find_all_dense_packed16(rows, slots, l)
            assert m0 == m1 and (c0 == c1).all()
            assert int(c0.sum()) > n // 10
            filed = torch.arange(slots, device="cuda")[None, :] < c0[:, None]
            assert ((se & 0xFFFF)[filed] == s0[filed]).all() and (((se >> 16) & 0xFFFF)[filed] == e0[filed]).all()
            assert (se[~filed] == -1).all()  # slots beyond the count stay untouched
    # host buffers: needle_find_all_packed16_host (unfiled slots come back as 0xFFFFFFFF)
    h = rows.cpu().numpy()
    hl = lens.cpu().numpy().astype(np.uint32)
    hc, hse, hm = p.find_all_dense_packed16(h[:3000], 40, hl[:3000])
    c1, se, m1 = p.find_all_dense_packed16(rows[:3000].contiguous(), 40, lens[:3000].contiguous())
    assert hm == m1 and (hc == c1.cpu().numpy().astype(np.uint32)).all() and (hse.view(np.int32) == se.cpu().numpy()).all()
    wide = torch.zeros((2, 65536 + 16), dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception):
        p.find_all_dense_packed16(wide, 4)
