"""UTF-16 rows behind the byte program's n-gram filter (needle_amd/csrc/needle_ngram.hip, CW = 2; needle_ngram.h narrow16): for patterns
whose chars all lie below 0xFF the filter kernel narrows the text as it loads it and runs the 8-bit program -- a char above 0xFE is "beyond
maxChar" (DFAClassBuilder.java:440, :565: `c > maxChar`), exactly what byte 0xFF is to the byte program.  The answers must be the CPU
oracle's on the UTF-16 rows, bit for bit, and those of the UTF-16 scan kernels with the route switched off (NEEDLE_PREFILTER_UTF16=0);
`filter_launches` of needle_pattern_prefilter_state proves which kernel ran.  Shapes: full rows, ragged rows, strides of 64 / 192 / 256 /
1024 chars, batches ending inside a group and inside a unit, keywords at both ends of a row, chars above 0xFF around and inside keywords'
places (CJK, 0x0100 + a keyword char: same low byte, another char), dictionaries on SEVERAL pages of the BMP behind the WIDE filter (windows
of four 16-bit code units: DFA.java:438-463 -- the reference's class map covers every code unit of any pattern), one-dword results, every match of every row (dense, one dword per match, counting pass, compact filing), the big dictionary whose walks
leave the LDS."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler, unpack_bitmap
from test_compile_matches_txt import oracle_for
route_on = int(sys.argv[1])
import os
wide_on = os.environ.get("NEEDLE_PREFILTER_WIDE", "1") != "0"
dev = "cuda"
rng = np.random.default_rng(7)

def utf16_rows(words, n, stride, seed):
    """keyword text as UTF-16, with chars above 0xFF sprinkled in: plain CJK, and `0x0100 | c` for chars c of the text itself
    (the low byte of a keyword char under a non-zero high byte must NOT read as that char)"""
    rows8 = W.keyword_batch(np, words, seed, n, stride)
    r = rows8.astype(np.uint16)
    g = np.random.default_rng(seed)
    m = g.random(r.shape) < 0.02
    r[m] = g.integers(0x4E00, 0x9FFF, size=int(m.sum()), dtype=np.uint16)
    m2 = g.random(r.shape) < 0.02
    r[m2] |= 0x0100
    m3 = g.random(r.shape) < 0.005
    r[m3] = 0x00FF   # the substitute itself, and 0xFF00-ish values
    m4 = g.random(r.shape) < 0.005
    r[m4] = 0xFF00 | (r[m4] & 0xFF)
    # keywords at both ends of some rows (row start: no window reaches back; row end: cut by the row)
    for i in range(0, n, 7):
        w = words[i % len(words)]
        r[i, :len(w)] = np.frombuffer(w.encode(), dtype=np.uint8)
        w2 = words[(i * 3 + 1) % len(words)]
        r[i, stride - len(w2):] = np.frombuffer(w2.encode(), dtype=np.uint8)
        if i % 14 == 0: r[i, stride - len(w2) + 1] |= 0x0100  # ... broken by a high byte
    return r

def check(p, o, host, lens, tag):
    n = host.shape[0]
    rows = torch.from_numpy(host.view(np.int16)).to(dev)
    dl = None if lens is None else torch.from_numpy(lens.astype(np.int32)).to(dev)
    before = p.prefilter_state("forwards")["filter_launches"] + p.prefilter_state("contained_in")["filter_launches"]
    fw, fs, fe = p.find_batch(rows, dl)
    cw = p.contained_in_batch(rows, dl)
    pw, pk = p.find_packed16_batch(rows, dl)
    torch.cuda.synchronize()
    after = p.prefilter_state("forwards")["filter_launches"] + p.prefilter_state("contained_in")["filter_launches"]
    hl = None if lens is None else lens.astype(np.uint32)
    of, ofs, ofe = o.batch_find(host, hl, threads=8)
    oc = o.batch_contained_in(host, hl, threads=8)
    got = unpack_bitmap(fw, n)
    bad = np.nonzero(got != of)[0]
    assert bad.size == 0, (tag, "find bitmap", bad[:5], n)
    fs, fe = fs.cpu().numpy(), fe.cpu().numpy()
    bs = np.nonzero((fs != ofs) | (fe != ofe))[0]
    assert bs.size == 0, (tag, "start/end", bs[:5], fs[bs[:5]], ofs[bs[:5]], fe[bs[:5]], ofe[bs[:5]])
    assert (unpack_bitmap(cw, n) == oc).all(), (tag, "containedIn")
    pkv = pk.cpu().numpy().view(np.uint32)
    assert (unpack_bitmap(pw, n) == of).all() and ((pkv & 0xFFFF)[of] == ofs[of]).all() and ((pkv >> 16)[of] == ofe[of]).all() and (pkv[~of] == 0xFFFFFFFF).all(), (tag, "packed16")
    # every match of every row through the filter kernel's find-all form (dense slots, one dword per match, the counting pass, the
    # compact filing) against the oracle's repeated find() on every 5th row
    want = {i: o.find_all(host[i] if hl is None else host[i, :hl[i]]) for i in range(0, n, 5)}
    slots = max([len(w) for w in want.values()] + [1]) + 1
    counts, st, en, more = p.find_all_dense(rows, slots, dl)
    c2, se, more2 = p.find_all_dense_packed16(rows, slots, dl)
    cnt = p.count_matches_batch(rows, dl).cpu().numpy()
    offs, s1, e1 = p.find_all_csr(rows, dl)
    torch.cuda.synchronize()
    counts, st, en = counts.cpu().numpy(), st.cpu().numpy(), en.cpu().numpy()
    for i, w in want.items():
        k = min(len(w), slots)
        assert counts[i] == k and list(zip(st[i, :k].tolist(), en[i, :k].tolist())) == w[:k], (tag, "find-all", i, counts[i], st[i], en[i], w[:6])
    assert (c2.cpu().numpy() == counts).all() and more2 == more, (tag, "find-all packed counts")
    sev = se.cpu().numpy().view(np.uint32)
    filed = np.arange(slots)[None, :] < counts[:, None]
    assert ((sev & 0xFFFF)[filed] == st[filed]).all() and ((sev >> 16)[filed] == en[filed]).all(), (tag, "find-all packed")
    assert (np.minimum(cnt, slots) == counts).all() and bool(more) == bool((cnt > slots).any()), (tag, "count pass")
    offs, s1, e1 = offs.cpu().numpy(), s1.cpu().numpy(), e1.cpu().numpy()
    assert (np.diff(offs) == cnt).all() and offs[-1] == cnt.sum() == len(s1), (tag, "csr offsets")
    fits = cnt <= slots
    csr_row = np.repeat(np.arange(n), cnt)
    assert (s1[fits[csr_row]] == st[filed & fits[:, None]]).all() and (e1[fits[csr_row]] == en[filed & fits[:, None]]).all(), (tag, "csr matches")
    final = p.prefilter_state("forwards")["filter_launches"] + p.prefilter_state("contained_in")["filter_launches"]
    return after - before, int(of.sum()), final - after

big = W.keywords(1000, min_len=6, max_len=8)
huge = W.keywords(3000, min_len=6, max_len=8)
small = ["Sherlock", "Holmes", "Watson", "Moriarty", "Mycroft", "Baskerville"]
total_launches = 0
for name, words, shapes in (("1000 keywords", big, [(20000, 256), (4099, 192), (2500, 1024), (7001, 64)]),
                            ("3000 keywords (walks out of L2)", huge, [(20000, 256), (3001, 192)]),
                            ("six names", small, [(9000, 256)])):
    rx = "|".join(words)
    p = DFACompiler.compile(rx, "t", 0)
    o, _ = oracle_for(rx, 0)
    for n, stride in shapes:
        host = utf16_rows(words, n, stride, 1000 + n)
        launches, hits, fa_launches = check(p, o, host, None, (name, n, stride, "full"))
        total_launches += launches
        assert hits > 0
        if name != "six names":
            assert (launches >= 3) == bool(route_on), (name, n, stride, launches)
            assert (fa_launches >= 4) == bool(route_on), (name, n, stride, "find-all launches", fa_launches)
        lens = rng.integers(0, stride + 1, size=n)
        lens[::5] = stride
        launches, _, _ = check(p, o, host, lens, (name, n, stride, "ragged"))
        total_launches += launches
# a dictionary on ANOTHER page of the BMP (Cyrillic: page 4): the byte program of the tables rebased to the page, every char outside the
# page -- the spaces between the words, ASCII, CJK, and chars of page 5 with a keyword char's low byte -- narrowed to the page's substitute
def cyr(w): return "".join(chr(0x0430 + ord(c) - 97) for c in w)
for name, words, shapes in (("1000 Cyrillic keywords", big, [(20000, 256), (4099, 192)]), ("3000 Cyrillic keywords (walks out of L2)", huge, [(12000, 256)])):
    cwords = [cyr(w) for w in words]
    rx = "|".join(cwords)
    p = DFACompiler.compile(rx, "t", 0)
    o, _ = oracle_for(rx, 0)
    for n, stride in shapes:
        host = utf16_rows(words, n, stride, 3000 + n)          # (the Latin text with its sprinkles ...)
        low = (host >= 97) & (host <= 122)
        host[low] += 0x0430 - 97                                # ... its letters moved to the Cyrillic page: the keywords are in it now
        g = np.random.default_rng(n)
        m = g.random(host.shape) < 0.01
        host[m] = g.integers(0x0400, 0x042F, size=int(m.sum()), dtype=np.uint16)   # page 4, not in the pattern (capitals)
        m = low & (g.random(host.shape) < 0.01)
        host[m] += 0x0100                                       # page 5, the low byte of a keyword char
        m = g.random(host.shape) < 0.01
        host[m] = g.integers(97, 123, size=int(m.sum()), dtype=np.uint16)          # page 0, the low byte range of nothing in particular
        launches, hits, fa_launches = check(p, o, host, None, (name, n, stride, "full"))
        total_launches += launches
        assert hits > 0
        assert (launches >= 3) == bool(route_on), (name, n, stride, launches)
        lens = rng.integers(0, stride + 1, size=n)
        launches, _, _ = check(p, o, host, lens, (name, n, stride, "ragged"))
        total_launches += launches
# a pattern on TWO pages (Latin and Cyrillic keywords) has no byte program -- it takes the WIDE filter (needle_ngram.h ngram_piece16: windows
# of four 16-bit code units hashed as they stand, candidates verified on the UTF-16 HBM-table program); NEEDLE_PREFILTER_WIDE=0: the UTF-16
# scan kernels
p = DFACompiler.compile("|".join(big[:200] + [cyr(w) for w in big[200:400]]), "t", 0)
o, _ = oracle_for("|".join(big[:200] + [cyr(w) for w in big[200:400]]), 0)
assert p.utf16_route() is None
host = utf16_rows(big[:200], 6000, 256, 77)
host[::2][(host[::2] >= 97) & (host[::2] <= 122)] += 0x0430 - 97
launches, hits, fa_launches = check(p, o, host, None, ("two pages",))
assert hits > 0 and (launches >= 3) == bool(route_on and wide_on) and (fa_launches >= 4) == bool(route_on and wide_on), (launches, fa_launches)
total_launches += launches
# three scripts on ~25 pages (Latin, Cyrillic, CJK: needle_amd.workload.keywords_mixed), mixed-script rows: full / ragged rows, strides of
# 64 / 192 / 256 / 1024 chars, batches ending inside a group, keywords of every script at both ends of rows, adjacent matches
for per, shapes in ((300, [(20000, 256), (4099, 192), (1500, 1024), (7001, 64)]), (1000, [(12000, 256)])):
    words = W.keywords_mixed(per)
    rx = "|".join(words)
    p = DFACompiler.compile(rx, "t", 0)
    o, _ = oracle_for(rx, 0)
    assert p.utf16_route() is None
    for n, stride in shapes:
        host = W.mixed_keyword_batch(np, words, 4000 + n, n, stride)
        for i in range(0, n, 9):
            w = [ord(c) for c in words[(i * 5) % len(words)]]
            w2 = [ord(c) for c in words[(i * 7 + 1) % len(words)]]
            host[i, :len(w)] = w
            host[i, stride - len(w2):] = w2
            if i % 18 == 0 and stride >= 64: host[i, 20:20 + len(w)] = w; host[i, 20 + len(w):20 + len(w) + len(w2)] = w2
            if i % 27 == 0: host[i, stride - 2] ^= 0x0100  # ... the keyword at the row's end broken
        launches, hits, fa_launches = check(p, o, host, None, ("mixed scripts", per, n, stride, "full"))
        total_launches += launches
        assert hits > 0 and (launches >= 3) == bool(route_on and wide_on) and (fa_launches >= 4) == bool(route_on and wide_on), (per, n, stride, launches, fa_launches)
        lens = rng.integers(0, stride + 1, size=n)
        lens[::5] = stride
        launches, _, _ = check(p, o, host, lens, ("mixed scripts", per, n, stride, "ragged"))
        total_launches += launches
# a pattern with a char at or above 0xFF on page 0 takes the route as long as the page has a char of the "other" class left
p = DFACompiler.compile("abcdefÿgh|bcdefgh", "t", 0)
o, _ = oracle_for("abcdefÿgh|bcdefgh", 0)
host = utf16_rows(["abcdefgh", "bcdefgh"], 5000, 256, 5)
host[::3, 10:19] = np.array([97, 98, 99, 100, 101, 102, 0xFF, 103, 104], dtype=np.uint16)
launches, hits, _ = check(p, o, host, None, ("char 0xFF in the pattern",))
assert hits > 0
print("OK", total_launches)
'''


def run_child(route_on, wide_on=True):
    env = dict(os.environ)
    env["NEEDLE_PREFILTER_UTF16"] = "1" if route_on else "0"
    env["NEEDLE_PREFILTER_WIDE"] = "1" if wide_on else "0"
    r = subprocess.run([sys.executable, "-c", CODE, str(route_on)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    return int(r.stdout.strip().split()[-1])


@pytest.mark.gpu
def test_utf16_rows_behind_the_byte_filter_match_the_oracle():
    assert run_child(1) > 0


@pytest.mark.gpu
def test_utf16_route_off_is_the_same_answer():
    assert run_child(0) == 0


@pytest.mark.gpu
def test_wide_filter_off_is_the_same_answer():
    """NEEDLE_PREFILTER_WIDE=0: multi-page dictionaries on the UTF-16 scan kernels (the asserts on launch counts flip inside the child)."""
    assert run_child(1, wide_on=False) > 0
