"""SURVEY.md s8 f-4 on the device: containedIn() / find() behind the n-gram candidate filter (needle_amd/csrc/needle_ngram.hip)
must give the reference's answers -- bit for bit what the CPU oracle (oracle/needle_walk.c: DFAClassBuilder.java:335-471,
625-659, 956-1025) computes, and what the ordinary scan kernel computes with the filter switched off.  NEEDLE_PREFILTER is read
once per process: 1 (default: automata in the compressed form), 2 (every LDS-table automaton that allows a filter), 0 (off)
each run in a child.  Shapes: full rows, ragged rows, rows 64 / 192 / 1024 bytes apart, batches that end inside a 64-row group
and inside a 1 KiB unit, keywords at both ends of a row and cut by it, near-miss text that floods the filter with candidates."""
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
level = int(sys.argv[1])
big = W.keywords(1000, min_len=6, max_len=8)
small = ["Sherlock", "Holmes", "Watson", "Moriarty", "Mycroft", "Baskerville"]
cases = [("|".join(big), big, 6, True)]
# patterns WITHOUT bounded match lengths behind the filter (find()'s starts by backward walks, needle_ngram.hip BWD): a dictionary followed by
# a run of digits -- the automaton in no LDS form (HBM-table filter program, get_program variant 12) ...
ub = [w + str(10 + 7 * i % 90) for i, w in enumerate(big)]
cases += [("(" + "|".join(big) + ")[0-9]+", ub, 3, True)]
if level == 2:  # ... and small ones as plain LDS tables
    names = [w + d for w, d in zip(small, ["1", "22", "333", "4", "55", "6"])]
    cases += [("(" + "|".join(small) + ")[0-9]+", names, None, True), ("(abcdef|bcdefgh|cdefghij)[xy]+", ["abcdefxy", "bcdefghxxxy", "cdefghijyyxxxy"], None, True)]
if level == 2:
    cases += [("|".join(small), small, None, True), ("abcdef|bcdefgh|cdefghij|xabcde", ["abcdef", "bcdefgh", "cdefghij", "xabcde"], None, True),
              ("abcdefgh", ["abcdefgh"], None, True), ("|".join(W.keywords(200, min_len=5, max_len=9)), W.keywords(200, min_len=5, max_len=9), None, True)]
dev = "cuda"
def check(p, o, rows, lens, tag):
    n = rows.shape[0]
    host = rows.cpu().numpy()
    hl = None if lens is None else lens.cpu().numpy().astype(np.uint32)
    fw, fs, fe = p.find_batch(rows, lens)
    cw = p.contained_in_batch(rows, lens)
    torch.cuda.synchronize()
    of, ofs, ofe = o.batch_find(host, hl, threads=8)
    oc = o.batch_contained_in(host, hl, threads=8)
    got = unpack_bitmap(fw, n)
    bad = np.nonzero(got != of)[0]
    assert bad.size == 0, (tag, "find bitmap", bad[:5], n)
    bs = np.nonzero((fs.cpu().numpy() != ofs) | (fe.cpu().numpy() != ofe))[0]
    assert bs.size == 0, (tag, "start/end", bs[:5], fs.cpu().numpy()[bs[:5]], ofs[bs[:5]], fe.cpu().numpy()[bs[:5]], ofe[bs[:5]])
    assert (unpack_bitmap(cw, n) == oc).all(), (tag, "containedIn")
    # every match of every row (the filter kernel's find-all form: needle_find_all_dev behind the filter) against the oracle's repeated
    # find() on every 3rd row, the one-dword form against the two arrays on all of them; 3 slots: rows with more set `more`
    want = {i: o.find_all(host[i] if hl is None else host[i, :hl[i]]) for i in range(0, n, 3)}
    most = max([len(w) for w in want.values()] + [1])
    for slots in (most + 1, 2):
        counts, st, en, more = p.find_all_dense(rows, slots, lens)
        c2, se, more2 = p.find_all_dense_packed16(rows, slots, lens)
        torch.cuda.synchronize()
        counts, st, en = counts.cpu().numpy(), st.cpu().numpy(), en.cpu().numpy()
        assert (c2.cpu().numpy() == counts).all() and more2 == more, (tag, "find-all packed counts")
        sev = se.cpu().numpy().view(np.uint32)
        filed = np.arange(slots)[None, :] < counts[:, None]
        assert ((sev & 0xFFFF)[filed] == st[filed]).all() and ((sev >> 16)[filed] == en[filed]).all(), (tag, "find-all packed")
        c3, blocks, more3 = p.find_all_blocked16(rows, slots, lens)  # group-blocked slots behind the filter
        bv = p.unblock16(blocks, n).cpu().numpy().view(np.uint32)
        assert (c3.cpu().numpy() == counts).all() and more3 == more and (bv[filed] == sev[filed]).all() and (bv[~filed] == 0xFFFFFFFF).all(), (tag, "find-all blocked")
        o2, se2, more4 = p.find_all_compact16(rows, slots, lens)  # the compact form in one call, behind the filter
        o2, se2 = o2.cpu().numpy(), se2.cpu().numpy().view(np.uint32)
        assert more4 == more and (np.diff(o2) == counts).all() and o2[-1] == counts.sum() == len(se2) and (se2 == sev[filed]).all(), (tag, "find-all compact16")
        for i, w in want.items():
            k = min(len(w), slots)
            assert counts[i] == k and list(zip(st[i, :k].tolist(), en[i, :k].tolist())) == w[:k], (tag, "find-all", i, counts[i], st[i], en[i], w[:6])
        if slots > most:
            assert counts[::3].sum() == sum(len(w) for w in want.values())
            # the counting pass and the compact (CSR) filing through the same kernel
            cnt = p.count_matches_batch(rows, lens).cpu().numpy()
            # (`most` is over the sampled rows: another row may have more matches than the dense form has slots -- it says so)
            assert (np.minimum(cnt, slots) == counts).all() and bool(more) == bool((cnt > slots).any()), (tag, "count pass")
            offs, s1, e1 = p.find_all_csr(rows, lens)
            offs, s1, e1 = offs.cpu().numpy(), s1.cpu().numpy(), e1.cpu().numpy()
            assert (np.diff(offs) == cnt).all() and offs[-1] == cnt.sum() == len(s1), (tag, "csr offsets")
            fits = cnt <= slots
            csr_row = np.repeat(np.arange(n), cnt)
            assert (s1[fits[csr_row]] == st[filed & fits[:, None]]).all() and (e1[fits[csr_row]] == en[filed & fits[:, None]]).all(), (tag, "csr matches")
    return int(of.sum())
for rx, words, mode, expect in cases:
    p = DFACompiler.compile(rx, "t", 0)
    o, _ = oracle_for(rx, 0)
    fi, ci = p.prefilter_info("forwards"), p.prefilter_info("contained_in")
    if level == 0:
        assert not fi["on"] and not ci["on"]
    else:
        assert fi["on"] and ci["on"], (rx[:40], fi, ci)
    if mode is not None and level > 0:
        assert fi["mode"] == mode, fi
    kw = [torch.tensor([ord(c) for c in w], dtype=torch.uint8, device=dev) for w in words[:8]]
    total = 0
    for stride, n in ((256, 64 * 700 + 13), (64, 64 * 900 + 7), (192, 64 * 300 + 63), (1024, 64 * 60 + 1), (256, 70),
                      (112, 64 * 500 + 9), (272, 64 * 200 + 33), (208, 64 * 150 + 1)):  # (groups that are not whole batches of units)
        rows = W.keyword_batch(torch, words, 3, n, stride, device=dev)
        k0, k1, k2 = kw[0], kw[1 % len(kw)], kw[2 % len(kw)]
        rows[::11, stride - len(k0):] = k0                  # a keyword that ends with the row
        rows[5::11, stride - len(k1) + 1:] = k1[:-1]        # one that the row's end cuts
        rows[7::11, :len(k2)] = k2                          # one at the very start
        rows[9::11, 1:1 + len(k0)] = k0                     # one a char in (the window reaching back over the row start)
        for j in range(0, stride - 12, 16):                 # keywords across every 16-byte piece boundary and 1 KiB unit boundary
            r = 13 + 11 * (j // 16)
            if r < n:
                rows[r, j + 12:j + 12 + min(len(k1), stride - j - 12)] = k1[:stride - j - 12]
        total += check(p, o, rows, None, (rx[:30], stride, n, "full"))
        lens = (torch.arange(n, device=dev, dtype=torch.int64) * 2654435761 % (stride + 1)).to(torch.int32)
        total += check(p, o, rows, lens, (rx[:30], stride, n, "ragged"))
    # near misses: rows made of keyword prefixes / keywords with one char changed -- candidates everywhere, few matches
    g = torch.Generator(device=dev); g.manual_seed(5)
    n = 64 * 200 + 5
    wt = torch.zeros((len(words), 16), dtype=torch.uint8, device=dev) + 32
    for i, w in enumerate(words):
        wt[i, :len(w)] = torch.tensor([ord(c) for c in w], dtype=torch.uint8, device=dev)
    pick = torch.randint(0, len(words), (n, 16), device=dev, generator=g)
    rows = wt[pick].reshape(n, 256).clone()
    flip = torch.rand((n, 256), device=dev, generator=g) < 0.08
    rows = torch.where(flip, torch.full_like(rows, ord("q")), rows)
    total += check(p, o, rows, None, (rx[:30], "near-miss"))
    assert total > 1000, total
    # a flood: rows built from the keywords' own TAILS (first char replaced) -- every slot passes the filter, almost nothing matches --
    # with real keywords planted in between: the kernel gives such groups up and walks their rows in full (needle_ngram.hip kNgFlood)
    long_words = [w for w in words if len(w) >= 6][:512] or words
    for stride, n in ((256, 64 * 150 + 9), (64, 64 * 200 + 3), (1024, 64 * 20 + 1)):
        wt8 = torch.zeros((len(long_words), 8), dtype=torch.uint8, device=dev) + 32
        for i, w in enumerate(long_words):
            t = torch.tensor([ord(c) for c in w[-8:]], dtype=torch.uint8, device=dev)
            t[0] = ord("q") if t[0] != ord("q") else ord("z")
            wt8[i, 8 - len(t):] = t
        pick = torch.randint(0, len(long_words), (n, stride // 8), device=dev, generator=g)
        rows = wt8[pick].reshape(n, stride).clone()
        rows[::5, 8:8 + len(kw[0])] = kw[0]
        rows[3::7, stride - len(kw[1 % len(kw)]):] = kw[1 % len(kw)]
        total += check(p, o, rows, None, (rx[:30], stride, "flood"))
        lens = (torch.arange(n, device=dev, dtype=torch.int64) * 2654435761 % (stride + 1)).to(torch.int32)
        total += check(p, o, rows, lens, (rx[:30], stride, "flood ragged"))
        half = rows.clone()
        half[n // 2:] = W.keyword_batch(torch, words, 9, n - n // 2, stride, device=dev)  # quiet groups behind flooded ones
        total += check(p, o, half, None, (rx[:30], stride, "flood then quiet"))
if level == 2:
    # ADVICE r4: a candidate run that CROSSES an earlier accept and lives on (the search automaton keeps the higher-priority longer
    # alternative, the restart threads are pruned) knows nothing about its window: `nation` right behind `inter` inside `internationa..`
    rx = "international|inter|nation|qrstuvwxyzab"
    p = DFACompiler.compile(rx, "t", 0)
    o, _ = oracle_for(rx, 0)
    assert p.prefilter_info("forwards")["on"]
    pieces = ["international", "inter", "nation", "internation", "internationa", "nationa", " ", "x", "qrstuvwxyzab", "tion", "internationwide"]
    rng = np.random.default_rng(11)
    for stride, n in ((256, 64 * 120 + 5), (64, 64 * 200 + 9)):
        host = np.full((n, stride), 32, dtype=np.uint8)
        for r in range(n):
            s_ = "".join(pieces[k] for k in rng.integers(0, len(pieces), size=stride // 3))[:stride]
            host[r, :len(s_)] = np.frombuffer(s_.encode(), dtype=np.uint8)
        rows = torch.from_numpy(host).to(dev)
        check(p, o, rows, None, ("crossing", stride))
        lens = (torch.arange(n, device=dev, dtype=torch.int64) * 2654435761 % (stride + 1)).to(torch.int32)
        check(p, o, rows, lens, ("crossing ragged", stride))
print("PREFILTER-GPU-OK")
'''


@pytest.mark.gpu
@pytest.mark.parametrize("level", [1, 2, 0], ids=["default", "all-table-automata", "off"])
def test_prefilter_levels_match_oracle(level):
    env = dict(os.environ, NEEDLE_PREFILTER=str(level))
    if level == 2:
        env["NEEDLE_PAIR_MAX_BYTES"] = "0"  # (the small automata as plain uint8 tables: the pair table has no filter kernel)
    r = subprocess.run([sys.executable, "-c", CODE, str(level)], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert "PREFILTER-GPU-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("min_len", [5, 7])
def test_two_sided_second_level_at_row_and_batch_ends(min_len):
    """The two-sided second level (NgramParams::on2 == 2: shortest match 5 -> stride 2, 7 -> stride 4) asks for the 5 chars ending one char
    BEHIND a candidate's window: at a row's last char there is none (the next row's first char, or -- the batch's last row -- nothing at all:
    the read stays inside the batch).  Keywords planted at the very end of rows, of the LAST rows of full and partial 64-row groups, cut by
    one char, and at the rows' starts; full rows and ragged ones; against the oracle on every row."""
    import numpy as np
    import torch
    from needle_amd import workload as W
    from needle_amd.pattern import unpack_bitmap
    from test_gpu_configs import compiled
    words = W.keywords(1000, min_len=min_len, max_len=min_len + 3)  # (big enough for the compressed automaton: the filter's default territory)
    p, o = compiled("|".join(words))
    i = p.prefilter_info("forwards")
    assert i["on"] == 1 and i["on2"] == 2 and i["stride"] == (2 if min_len == 5 else 4) and i["min_len"] == min_len, i
    enc = [np.frombuffer(w.encode(), dtype=np.uint8) for w in words]
    for width in (64, 256, 320):
        for n in (64 * 300, 64 * 300 + 1, 70_000 + 37):
            host = W.keyword_batch(np, words, 3, n, width).copy()
            for j, r in enumerate(range(n - 1, max(n - 200, -1), -1)):
                w = enc[j % len(enc)]
                if j % 4 == 0: host[r, width - len(w):] = w                      # ends with the row
                elif j % 4 == 1: host[r, width - len(w) + 1:] = w[:-1]            # cut by the row's end
                elif j % 4 == 2: host[r, :len(w)] = w                             # at the row's start
                else: host[r, width - len(w) - 1:width - 1] = w                   # one char in front of the end
            for lens in (None, ((np.arange(n, dtype=np.uint64) * 2654435761) % (width + 1)).astype(np.uint32)):
                rows = torch.from_numpy(host).cuda()
                tl = None if lens is None else torch.from_numpy(lens.astype(np.int32)).cuda()
                fw, fs, fe = p.find_batch(rows, tl)
                cw = p.contained_in_batch(rows, tl)
                torch.cuda.synchronize()
                m, os_, oe = o.batch_find(host, lens, threads=8)
                assert (unpack_bitmap(fw, n) == m).all() and (unpack_bitmap(cw, n) == m).all(), (min_len, width, n, lens is None)
                assert (fs.cpu().numpy() == os_).all() and (fe.cpu().numpy() == oe).all(), (min_len, width, n, lens is None)
                assert m[-200:].sum() >= 50 or lens is not None
