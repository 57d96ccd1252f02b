"""SURVEY.md s8f-1 on the device, in LOCK-STEP: needle_find_all_* / needle_count_matches_dev through the find-all transducer
(needle_amd/csrc/needle_find_all_ls.hip, needle_lower.h) against the CPU oracle's repeated find() (oracle/walker.py find_all: the
reference's Matcher.find() with its nextStart cursor, DFAClassBuilder.java:616-659) -- bit-exact counts, starts, ends and `more`
flag, in every result form (two arrays, one dword per match, counting pass, compact filing), on full and ragged rows, strides that
are and are not whole tiles, batches that end inside a 64-row group, 8- and 16-bit rows, matches pending at the row's end.
NEEDLE_FIND_ALL_LOCKSTEP is read once per process: the same checks run with it off (the per-lane one-pass kernel) in a child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler
from test_compile_matches_txt import oracle_for
lockstep = int(sys.argv[1])
dev = "cuda"
def check(p, o, rows, lens, tag, every=1):
    n = rows.shape[0]
    host = rows.cpu().numpy()
    if host.dtype == np.int16:
        host = host.view(np.uint16)
    hl = None if lens is None else lens.cpu().numpy()
    idx = list(range(0, n, every))
    want = {i: o.find_all(host[i] if hl is None else host[i, :hl[i]]) for i in idx}
    most = max([len(w) for w in want.values()] + [1])
    total = 0
    for slots in (most + 1, 2):
        counts, st, en, more = p.find_all_dense(rows, slots, lens)
        c2, se, more2 = p.find_all_dense_packed16(rows, slots, lens)
        torch.cuda.synchronize()
        counts, st, en = counts.cpu().numpy(), st.cpu().numpy(), en.cpu().numpy()
        assert (c2.cpu().numpy() == counts).all() and more2 == more, (tag, "packed counts")
        sev = se.cpu().numpy().view(np.uint32)
        filed = np.arange(slots)[None, :] < counts[:, None]
        assert ((sev & 0xFFFF)[filed] == st[filed]).all() and ((sev >> 16)[filed] == en[filed]).all(), (tag, "packed form")
        assert (st[~filed] == -1).all() and (en[~filed] == -1).all(), (tag, "slots beyond the count are untouched")
        c3, blocks, more3 = p.find_all_blocked16(rows, slots, lens)  # group-blocked slots: the same matches at [r >> 6, k, r & 63]
        bv = p.unblock16(blocks, n).cpu().numpy().view(np.uint32)
        assert (c3.cpu().numpy() == counts).all() and more3 == more, (tag, "blocked counts")
        assert (bv[filed] == sev[filed]).all() and (bv[~filed] == 0xFFFFFFFF).all(), (tag, "blocked form")
        assert (blocks.cpu().numpy().reshape(-1, slots, 64).transpose(0, 2, 1).reshape(-1, slots)[n:] == -1).all(), (tag, "rows beyond the batch")
        # the compact form in one call (needle_find_all_compact16_dev: the blocked pass + scan + compaction): offsets + a dense match array
        o2, se2, more4 = p.find_all_compact16(rows, slots, lens, cap=max(1, int(counts.sum()) // 2))  # (a capacity that is too small: retried)
        o2, se2 = o2.cpu().numpy(), se2.cpu().numpy().view(np.uint32)
        assert more4 == more and (np.diff(o2) == counts).all() and o2[0] == 0 and o2[-1] == counts.sum() == len(se2), (tag, "compact16 offsets")
        assert (se2 == sev[filed]).all(), (tag, "compact16 matches")
        for i, w in want.items():
            k = min(len(w), slots)
            assert counts[i] == k and list(zip(st[i, :k].tolist(), en[i, :k].tolist())) == w[:k], (tag, slots, i, counts[i], st[i].tolist(), en[i].tolist(), w[:8])
        if every == 1:
            assert bool(more) == (most > slots), (tag, more, most, slots)
        if slots > most:
            total = int(counts.sum())
            cnt = p.count_matches_batch(rows, lens).cpu().numpy()
            # (`most` is over the sampled rows: another row may have more matches than the dense form has slots -- it says so)
            assert (np.minimum(cnt, slots) == counts).all() and bool(more) == bool((cnt > slots).any()), (tag, "count pass")
            offs, s1, e1 = p.find_all_csr(rows, lens)
            offs, s1, e1 = offs.cpu().numpy(), s1.cpu().numpy(), e1.cpu().numpy()
            assert (np.diff(offs) == cnt).all() and offs[-1] == cnt.sum() == len(s1), (tag, "csr offsets")
            fits = cnt <= slots
            csr_row = np.repeat(np.arange(n), cnt)
            assert (s1[fits[csr_row]] == st[filed & fits[:, None]]).all() and (e1[fits[csr_row]] == en[filed & fits[:, None]]).all(), (tag, "csr matches")
    return total
total = 0
# ---- the bench dictionary (C3: 1000 keywords of 3..5 chars; the transducer is 1464 states in window layout)
words = W.keywords(1000)
rx = "|".join(words)
p = DFACompiler.compile(rx, "t", 0)
o, _ = oracle_for(rx, 0)
assert p.find_all_transducer(1) is not None and p.find_all_transducer(2) is not None
kw = [torch.tensor([ord(c) for c in w], dtype=torch.uint8, device=dev) for w in words[:4]]
for stride, n in ((256, 64 * 40 + 13), (64, 64 * 50 + 7), (192, 64 * 20 + 63), (1040, 64 * 6 + 1), (256, 70), (16, 64 * 30 + 5), (80, 64 * 9)):
    rows = W.keyword_batch(torch, words, 3, n, stride, device=dev)
    rows[::11, stride - len(kw[0]):] = kw[0]          # a keyword that ends with the row: pending at the row's end
    rows[5::11, stride - len(kw[1]) + 1:] = kw[1][:-1]  # one the row's end cuts
    rows[7::11, :len(kw[2])] = kw[2]                    # one at the very start
    total += check(p, o, rows, None, ("c3", stride, n, "full"), every=3)
    lens = (torch.arange(n, device=dev, dtype=torch.int64) * 2654435761 % (stride + 1)).to(torch.int32)
    total += check(p, o, rows, lens, ("c3", stride, n, "ragged"), every=3)
    if stride >= 64:
        total += check(p, o, rows.to(torch.int16), None, ("c3 utf16", stride, n), every=7)
assert total > 20000, total
# ---- small patterns: several match lengths, matches that stay pending while a longer alternative lives (codes with k > 0),
# adjacent and overlapping candidates, one-length patterns; random text over the pattern's own letters
rng = np.random.default_rng(3)
for rx in ["abc|bcd|cdefg|a|xyzzy|zzy", "(foo|foobar|bar|barbaz|baz)x?", "ab|abcd|cdx|dxyz", "Sherlock|Holmes|Watson|Irene|Adler|John|Baker",
           "abcdef|bcd|cdefgh|f", "Sherlock", "aab|ab|b", "[ab]c|a[bc]d|[abc]{4}", "abcdefgh|abcd", "a.c|ab"]:
    p = DFACompiler.compile(rx, "t", 0)
    o, _ = oracle_for(rx, 0)
    assert p.find_all_transducer(1) is not None, rx
    alpha = np.array(sorted(set(ord(c) for c in rx if c.isalnum())) + [32, 10, 200], dtype=np.uint8)
    for stride, n in ((128, 64 * 6 + 3), (48, 64 * 4 + 9), (320, 130)):
        host = alpha[rng.integers(0, len(alpha), size=(n, stride))]
        rows = torch.from_numpy(host).to(dev)
        total += check(p, o, rows, None, (rx, stride, "full"), every=2)
        lens = torch.from_numpy(rng.integers(0, stride + 1, size=n).astype(np.int32)).to(dev)
        total += check(p, o, rows, lens, (rx, stride, "ragged"), every=2)
        if stride != 48:
            total += check(p, o, rows.to(torch.int16), lens, (rx, stride, "utf16 ragged"), every=5)
# ---- patterns WITHOUT bounded match lengths whose matches are runs (BASELINE's C2 / C5 kind): the RUN transducer -- lock-step, the start
# of a match from the lane's run-start register instead of indexBackwards (needle_lower.h lower_find_all_runs); NEEDLE_FIND_ALL_RUNS=0:
# the one-pass kernel and its backward walks
for rx in ["[0-9]+", "[a-c]{3}[a-c]*", "a+b+", "ab*", "[0-9]+x", "[a-z]+[0-9]"]:
    p = DFACompiler.compile(rx, "t", 0)
    o, _ = oracle_for(rx, 0)
    ft = p.find_all_transducer(1)
    assert ft is not None and ft["kind"] == 2, rx
    alpha = np.array(sorted(set(ord(c) for c in rx if c.isalnum())) + [ord(c) for c in "0123456789abcx"] + [32, 32, 32, 10, 200], dtype=np.uint8)
    for stride, n in ((256, 64 * 9 + 3), (48, 64 * 4 + 9), (320, 130), (16, 64 * 11 + 1)):
        host = alpha[rng.integers(0, len(alpha), size=(n, stride))]
        host[::5, stride - 3:] = alpha[0]   # runs that end with the row
        rows = torch.from_numpy(host).to(dev)
        total += check(p, o, rows, None, (rx, stride, "runs full"), every=2)
        lens = torch.from_numpy(rng.integers(0, stride + 1, size=n).astype(np.int32)).to(dev)
        total += check(p, o, rows, lens, (rx, stride, "runs ragged"), every=2)
        if stride >= 48:
            total += check(p, o, rows.to(torch.int16), lens, (rx, stride, "runs utf16 ragged"), every=5)
# runs LONGER than 16 bits can hold, on rows beyond the one-dword forms' 65 535 chars (two int32 arrays; rows of up to 8 MB stay lock-step)
p = DFACompiler.compile("[0-9]+", "t", 0)
o, _ = oracle_for("[0-9]+", 0)
host = np.full((70, 70016), ord("a"), dtype=np.uint8)
host[::2, 10:66000] = ord("7")
host[1::2, 5:9] = ord("3")
host[1::2, 69000:] = ord("5")
host[3::4, 30000:30001] = ord("9")
cnt_, st_, en_, more_ = p.find_all_dense(torch.from_numpy(host).to(dev), 4)
cnt_, st_, en_ = cnt_.cpu().numpy(), st_.cpu().numpy(), en_.cpu().numpy()
for i in range(70):
    w = o.find_all(host[i])
    assert not more_ and cnt_[i] == len(w) and list(zip(st_[i, :len(w)].tolist(), en_[i, :len(w)].tolist())) == [tuple(x) for x in w], ("long runs", i, w)
rx = W.script_regex()  # C5: a run of >= 3 chars of 42 BMP ranges, UTF-16 rows
p = DFACompiler.compile(rx, "t", 0)
o, _ = oracle_for(rx, 0)
assert p.find_all_transducer(2) is not None and p.find_all_transducer(2)["kind"] == 2
for n in (64 * 30 + 7, 100):
    rows = W.script_batch(torch, 17, n, 256, device=dev)
    total += check(p, o, rows, None, ("c5", n, "full"), every=3)
    lens = (torch.arange(n, device=dev, dtype=torch.int64) * 2654435761 % 257).to(torch.int32)
    total += check(p, o, rows, lens, ("c5", n, "ragged"), every=3)
# ---- a pattern WITHOUT a transducer keeps the one-pass kernel whatever the switch says
rx = "international|inter|nation|qrstuvwxyzab"
p = DFACompiler.compile(rx, "t", 0)
o, _ = oracle_for(rx, 0)
assert p.find_all_transducer(1) is None
pieces = ["international", "inter", "nation", "internationa", " ", "x", "tion"]
host = np.full((300, 128), 32, dtype=np.uint8)
for r in range(300):
    s_ = "".join(pieces[k] for k in rng.integers(0, len(pieces), size=40))[:128]
    host[r, :len(s_)] = np.frombuffer(s_.encode(), dtype=np.uint8)
total += check(p, o, torch.from_numpy(host).to(dev), None, ("no transducer",))
print("LOCKSTEP-FIND-ALL-OK", lockstep, total)
'''


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"NEEDLE_FIND_ALL_LOCKSTEP": "0"}, {"NEEDLE_WINDOW": "0"}, {"NEEDLE_FIND_ALL_SHAPE": "8x64"}, {"NEEDLE_FIND_ALL_RUNS": "0"}],
                         ids=["lock-step", "off: one-pass kernel", "lock-step, column maps", "lock-step, 8 waves", "run patterns on the one-pass kernel"])
def test_find_all_every_form_vs_oracle(env):
    on = env.get("NEEDLE_FIND_ALL_LOCKSTEP", "1")
    r = subprocess.run([sys.executable, "-c", CODE, on], env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert "LOCKSTEP-FIND-ALL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
