"""Parity at BASELINE.json's FULL sizes (10^7 rows x 256 chars per GPU), where the CPU oracle would take minutes:
size-independent properties of the domain, each computed a second way with plain torch integer ops on the same
resident batch, plus the oracle itself on a seeded sample of rows spread over the whole batch.

  C2  [0-9]+      containedIn  <=>  the row holds a digit                 (torch compare + any)
                  find start   ==   index of the first digit, end == end of that digit run   (torch argmax)
                  matches      <=>  every char of the row is a digit
  C3  keywords    find.matched ==   containedIn bitmap; [start, end) spells one of the 1000 keywords; nothing matches
                  when the row is cut one char before `end` (lengths = end - 1): leftmost match really ends at `end`
  C5  BMP runs    find.matched ==   containedIn; [start, end) is a maximal run of in-range chars of length >= 3
  C5w scripts     find.matched ==   containedIn; [start, end) spells one alternative (Python `re.fullmatch` on sampled rows) and
                  `re.search` finds the same span; the span starts with a char of an alternative's first class
  C3  find-all    (every match of every row, 46.6 M of them) slot 0 == find(); every match spells a keyword; matches are
                  ordered and disjoint; count pass == filed counts; compact form == dense form; rounds of find_next agree
                  on the second match; the oracle's repeated find() on sampled rows
  all             partition invariance: the bitmap of the whole batch == the bitmaps of two unequal shards, joined
                  (what row sharding across GPUs relies on); a second launch gives identical bits"""
import numpy as np
import pytest

N_FULL = 10_000_000


def make(workload, n=N_FULL):
    import bench
    pattern, _what, words = bench.make_pattern(workload)
    rows = bench.make_rows(workload, words, 0, n, "cuda")
    return pattern, rows, words


def oracle_of(pattern):
    from oracle.walker import Dfa, OraclePattern
    t = pattern.tables()
    d = {k: Dfa(t["class_map"], t["stride"], v["table"], v["accepting"], v["max_char"]) for k, v in t["dfas"].items()}
    return OraclePattern(d["matches"], d["contained_in"], d["forwards"], d["backwards"], t["fixed_len"], -1)


def sample_rows(n, k=60_000, seed=7):
    rng = np.random.default_rng(seed)
    idx = np.unique(np.concatenate([rng.integers(0, n, k), np.arange(0, 4096), np.arange(n - 4096, n)]))
    return idx


def check_sample_against_oracle(pattern, rows, fw_bits, fs, fe, c_bits):
    import torch
    idx = sample_rows(rows.shape[0])
    host = rows[torch.from_numpy(idx).cuda()].cpu().numpy()
    if host.dtype == np.int16:
        host = host.view(np.uint16)
    o = oracle_of(pattern)
    of, ofs, ofe = o.batch_find(host, threads=8)
    assert (fw_bits[idx] == of).all()
    assert (fs[idx] == ofs).all() and (fe[idx] == ofe).all()
    assert (c_bits[idx] == o.batch_contained_in(host, threads=8)).all()


def partition_invariant(op, rows, whole_words):
    import torch
    cut = 64 * 61_237  # shards start on 64-row boundaries (needle_amd/sharding.py)
    a, b = op(rows[:cut]), op(rows[cut:])
    a, b = (a[0], b[0]) if isinstance(a, tuple) else (a, b)
    assert torch.equal(torch.cat([a, b]), whole_words)


@pytest.mark.gpu
def test_c2_full_size_properties():
    import torch
    from needle_amd.pattern import unpack_bitmap
    p, rows, _ = make("c2")
    n = rows.shape[0]
    cw = p.contained_in_batch(rows)
    again = p.contained_in_batch(rows)
    assert torch.equal(cw, again)
    c_bits = unpack_bitmap(cw, n)
    is_digit = (rows >= 48) & (rows <= 57)
    has_digit = is_digit.any(dim=1)
    assert (c_bits == has_digit.cpu().numpy()).all()
    assert abs(c_bits.mean() - 0.5) < 0.001
    fw, fs, fe = p.find_batch(rows)
    f_bits = unpack_bitmap(fw, n)
    assert (f_bits == c_bits).all()
    first = is_digit.to(torch.uint8).argmax(dim=1).to(torch.int32)  # first maximum == first digit
    want_start = torch.where(has_digit, first, torch.full_like(first, -1))
    assert torch.equal(fs, want_start)
    # end of the run that starts at `first`: first non-digit at or after it (or the row end)
    cols = torch.arange(256, device="cuda", dtype=torch.int32)[None, :]
    stop = (~is_digit) & (cols > first[:, None])
    has_stop = stop.any(dim=1)
    run_end = torch.where(has_stop, stop.to(torch.uint8).argmax(dim=1).to(torch.int32), torch.full_like(first, 256))
    assert torch.equal(fe, torch.where(has_digit, run_end, torch.full_like(first, -1)))
    del stop, cols
    m_bits = unpack_bitmap(p.matches_batch(rows), n)
    assert (m_bits == is_digit.all(dim=1).cpu().numpy()).all()
    partition_invariant(p.contained_in_batch, rows, cw)
    check_sample_against_oracle(p, rows, f_bits, fs.cpu().numpy(), fe.cpu().numpy(), c_bits)


@pytest.mark.gpu
def test_c3_full_size_properties():
    import torch
    from needle_amd.pattern import unpack_bitmap
    p, rows, words = make("c3")
    n = rows.shape[0]
    fw, fs, fe = p.find_batch(rows)
    fw2, fs2, fe2 = p.find_batch(rows)
    assert torch.equal(fw, fw2) and torch.equal(fs, fs2) and torch.equal(fe, fe2)
    f_bits = unpack_bitmap(fw, n)
    cw = p.contained_in_batch(rows)
    c_bits = unpack_bitmap(cw, n)
    assert (f_bits == c_bits).all()
    matched = torch.from_numpy(f_bits).cuda()
    assert bool(((fs >= 0) == matched).all()) and bool(((fe > fs) | ~matched).all())
    # [start, end) spells a keyword: base-32 code of the <= 5 letters, looked up in the keyword code set
    ln = (fe - fs).clamp(min=0)
    assert bool(((ln >= 3) & (ln <= 5) | ~matched).all())
    code = torch.zeros(n, dtype=torch.int64, device="cuda")
    for j in range(5):
        ch = rows.gather(1, (fs.long().clamp(min=0) + j).clamp(max=255)[:, None])[:, 0].long() - 96
        code = torch.where(ln > j, code * 32 + ch, code)
    wcodes = []
    for w in words:
        v = 0
        for ch in w:
            v = v * 32 + (ord(ch) - 96)
        wcodes.append(v)
    ok = torch.isin(code, torch.tensor(sorted(wcodes), dtype=torch.int64, device="cuda"))
    assert bool((ok | ~matched).all())
    # leftmost: with the row cut one char before `end`, indexForwards can no longer end at `end`; the match found
    # there (if any) must not start before `start`
    cut = torch.where(matched, fe - 1, torch.zeros_like(fe))
    _w, cs, _ce = p.find_batch(rows, cut)
    assert bool(((cs == -1) | (cs >= fs) | ~matched).all())
    partition_invariant(p.find_batch, rows, fw)
    check_sample_against_oracle(p, rows, f_bits, fs.cpu().numpy(), fe.cpu().numpy(), c_bits)


@pytest.mark.gpu
def test_c3_full_size_find_all_properties():
    """SURVEY.md s8f-1 at the full size: the one-pass find-all kernel over all 10^7 rows, tied to find() and to itself
    through properties that need no CPU pass, plus the oracle on a sample."""
    import torch
    from needle_amd.pattern import unpack_bitmap
    p, rows, words = make("c3")
    n = rows.shape[0]
    slots = 32
    counts, st, en, more = p.find_all_dense(rows, slots)
    assert not more
    fw, fs, fe = p.find_batch(rows)
    matched = torch.from_numpy(unpack_bitmap(fw, n)).cuda()
    # the first match IS find()
    assert bool(((counts > 0) == matched).all())
    assert torch.equal(torch.where(matched, st[:, 0], torch.full_like(fs, -1)), fs)
    assert torch.equal(torch.where(matched, en[:, 0], torch.full_like(fe, -1)), fe)
    # slots: filled up to the count, untouched (-1) beyond; matches ordered, disjoint, 3..5 chars long
    k = torch.arange(slots, device="cuda", dtype=torch.int32)[None, :]
    used = k < counts[:, None]
    assert bool(((st >= 0) == used).all()) and bool(((en >= 0) == used).all())
    ln = en - st
    assert bool((((ln >= 3) & (ln <= 5)) | ~used).all())
    assert bool(((st[:, 1:] >= en[:, :-1]) | ~used[:, 1:]).all())
    assert bool((en <= 256).all())
    total = int(counts.sum().item())
    assert total > 4 * n
    # every match spells a keyword (base-32 code of its letters in the keyword code set), checked slot by slot
    wcodes = []
    for w in words:
        v = 0
        for ch in w:
            v = v * 32 + (ord(ch) - 96)
        wcodes.append(v)
    wset = torch.tensor(sorted(wcodes), dtype=torch.int64, device="cuda")
    for slot in range(int(counts.max().item())):
        s_k, l_k, u_k = st[:, slot].long().clamp(min=0), ln[:, slot], used[:, slot]
        code = torch.zeros(n, dtype=torch.int64, device="cuda")
        for j in range(5):
            ch = rows.gather(1, (s_k + j).clamp(max=255)[:, None])[:, 0].long() - 96
            code = torch.where(l_k > j, code * 32 + ch, code)
        assert bool((torch.isin(code, wset) | ~u_k).all()), slot
    del code, ch
    # the second match == find() searched from the first one's end (the round-per-match form's second round)
    cursor = torch.where(matched, fe, torch.full_like(fe, -1))
    _w2, s2, e2 = p.find_next_batch(rows, cursor)
    has2 = counts > 1
    assert torch.equal(torch.where(has2, st[:, 1], torch.full_like(s2, -1)), s2)
    assert torch.equal(torch.where(has2, en[:, 1], torch.full_like(e2, -1)), e2)
    # count pass and compact form
    assert torch.equal(p.count_matches_batch(rows), counts)
    offsets, cs, ce = p.find_all_csr(rows)
    assert int(offsets[-1].item()) == total
    assert torch.equal(cs, st[used]) and torch.equal(ce, en[used])
    # run-to-run identity
    c2, st2, en2, _ = p.find_all_dense(rows, slots)
    assert torch.equal(c2, counts) and torch.equal(st2, st) and torch.equal(en2, en)
    # the oracle's repeated find() on rows spread over the batch
    o = oracle_of(p)
    idx = sample_rows(n, k=3000)
    host = rows[torch.from_numpy(idx).cuda()].cpu().numpy()
    hc, hs, he = counts[idx].cpu().numpy(), st[idx].cpu().numpy(), en[idx].cpu().numpy()
    for i in range(0, len(idx), 3):
        want = o.find_all(host[i])
        assert [(int(hs[i, j]), int(he[i, j])) for j in range(int(hc[i]))] == want, idx[i]


@pytest.mark.gpu
def test_c5_full_size_properties():
    import torch
    from needle_amd import workload as W
    from needle_amd.pattern import unpack_bitmap
    p, rows, _ = make("c5")
    n = rows.shape[0]
    fw, fs, fe = p.find_batch(rows)
    f_bits = unpack_bitmap(fw, n)
    cw = p.contained_in_batch(rows)
    c_bits = unpack_bitmap(cw, n)
    assert (f_bits == c_bits).all()
    # class membership a second way: a 65536-entry torch lookup built from the same explicit ranges
    lut = torch.zeros(65536, dtype=torch.bool, device="cuda")
    for a, b in W.SCRIPT_RANGES:
        lut[a:b + 1] = True
    matched = torch.from_numpy(f_bits).cuda()
    slab = 1 << 20
    for s in range(0, n, slab):
        r = rows[s:s + slab].long() & 0xFFFF
        member = lut[r]
        st, en, mt = fs[s:s + slab].long(), fe[s:s + slab].long(), matched[s:s + slab]
        cols = torch.arange(256, device="cuda")[None, :]
        inside = (cols >= st[:, None]) & (cols < en[:, None])
        assert bool(((member | ~inside).all(dim=1) | ~mt).all())  # every char of the span is in-range
        assert bool((((en - st) >= 3) | ~mt).all())
        after = member.gather(1, en.clamp(0, 255)[:, None])[:, 0] & (en < 256)
        assert bool((~after | ~mt).all())  # greedy: the run cannot be extended
        before = member.gather(1, (st - 1).clamp(0, 255)[:, None])[:, 0] & (st > 0)
        assert bool((~before | ~mt).all())  # leftmost: it is not the tail of a longer run
        # no match <=> no run of 3 in-range chars anywhere in the row
        run3 = member[:, :-2] & member[:, 1:-1] & member[:, 2:]
        assert bool((run3.any(dim=1) == mt).all())
    partition_invariant(p.find_batch, rows, fw)
    check_sample_against_oracle(p, rows, f_bits, fs.cpu().numpy(), fe.cpu().numpy(), c_bits)


@pytest.mark.gpu
def test_c5w_full_size_properties():
    """C5's wide variant at 10^7 UTF-16 rows (30 classes, 33 states: page-map lookups into an LDS table)."""
    import re
    import torch
    from needle_amd import workload as W
    from needle_amd.pattern import unpack_bitmap
    p, rows, _ = make("c5w")
    n = rows.shape[0]
    fw, fs, fe = p.find_batch(rows)
    f_bits = unpack_bitmap(fw, n)
    cw = p.contained_in_batch(rows)
    c_bits = unpack_bitmap(cw, n)
    assert (f_bits == c_bits).all()
    assert 0.25 < f_bits.mean() < 0.6
    # a match starts with a char of some alternative's first class and is at least 2 chars long (a second way: torch lookups)
    first = torch.zeros(65536, dtype=torch.bool, device="cuda")
    for _, els in W.SEQ_ALTS:
        first[els[0][0]:els[0][1] + 1] = True
    matched = torch.from_numpy(f_bits).cuda()
    slab = 1 << 20
    for s in range(0, n, slab):
        r = rows[s:s + slab].long() & 0xFFFF
        st, en, mt = fs[s:s + slab].long(), fe[s:s + slab].long(), matched[s:s + slab]
        c0 = r.gather(1, st.clamp(0, 255)[:, None])[:, 0]
        assert bool((first[c0] | ~mt).all())
        assert bool((((en - st) >= 2) & (en <= 256) | ~mt).all())
        assert bool((((st == -1) & (en == -1)) | mt).all())
    partition_invariant(p.find_batch, rows, fw)
    hs, he = fs.cpu().numpy(), fe.cpu().numpy()
    check_sample_against_oracle(p, rows, f_bits, hs, he, c_bits)
    # Python's own regex engine on a sample spread over the batch
    cre = re.compile(W.scriptseq_regex())
    idx = sample_rows(n, k=3000, seed=11)[::3]
    host = rows[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint16)
    for j, i in enumerate(idx):
        mm = cre.search("".join(map(chr, host[j])))
        assert ((True, mm.start(), mm.end()) if mm else (False, -1, -1)) == (bool(f_bits[i]), int(hs[i]), int(he[i])), i


@pytest.mark.gpu
def test_c3_full_size_find_all_with_the_lock_step_kernel_off():
    """The same find-all properties with NEEDLE_FIND_ALL_LOCKSTEP=0 (read once per process: a child): the per-lane one-pass kernel
    (needle_find_all.hip) stays correct at the full size beside the lock-step one that is the default for this dictionary."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_full_size.py", "-x", "-q", "-m", "gpu", "-k", "test_c3_full_size_find_all_properties"],
                       env=dict(os.environ, NEEDLE_FIND_ALL_LOCKSTEP="0"), capture_output=True, text=True, timeout=1800, cwd=root)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["c3s", "c3x"])
def test_c3s_find_all_behind_the_filter_full_size(workload):
    """C3-sparse (and c3x: 3000 keywords at the reference's state limit, the filter's walks out of HBM / L2) at 10^7 rows: find() and
    containedIn() against the oracle on rows sampled over the whole batch; every match of every row through the n-gram filter kernel's
    find-all form: slot 0 == find() on every row (another kernel path), counts == the counting pass, matches ordered and disjoint, every
    match spells a keyword, the oracle's repeated find() on sampled rows."""
    import torch
    from needle_amd.pattern import unpack_bitmap
    p, rows, words = make(workload)
    assert p.prefilter_info("forwards")["on"] == 1 and p.prefilter_info("forwards")["mode"] == (6 if workload == "c3s" else 3)
    if workload == "c3x":
        fw0, fs0, fe0 = p.find_batch(rows)
        cw0 = p.contained_in_batch(rows)
        assert torch.equal(fw0, cw0)
        f_bits = unpack_bitmap(fw0, rows.shape[0])
        assert 0.24 < f_bits.mean() < 0.26
        check_sample_against_oracle(p, rows, f_bits, fs0.cpu().numpy(), fe0.cpu().numpy(), unpack_bitmap(cw0, rows.shape[0]))
        partition_invariant(p.find_batch, rows, fw0)
    n, slots = rows.shape[0], 4
    counts, st, en, more = p.find_all_dense(rows, slots)
    assert not more
    fw, fs, fe = p.find_batch(rows)
    matched = torch.from_numpy(unpack_bitmap(fw, n)).cuda()
    assert bool(((counts > 0) == matched).all())
    assert bool((st[:, 0] == fs).all()) and bool((en[:, 0] == fe).all())
    assert bool((p.count_matches_batch(rows) == counts).all())
    filed = torch.arange(slots, device="cuda")[None, :] < counts[:, None]
    assert bool(((en > st) | ~filed).all()) and bool(((st[:, 1:] >= en[:, :-1]) | ~filed[:, 1:]).all())
    assert bool((st[~filed] == -1).all())
    kw = set(words)
    idx = sample_rows(n, k=4000, seed=23)
    hc, hs, he = counts[idx].cpu().numpy(), st[idx].cpu().numpy(), en[idx].cpu().numpy()
    host = rows[torch.from_numpy(idx).cuda()].cpu().numpy()
    o = oracle_of(p)
    total = 0
    for i in range(len(idx)):
        got = [(int(hs[i, j]), int(he[i, j])) for j in range(int(hc[i]))]
        for a, b in got:
            assert bytes(host[i, a:b]).decode() in kw
        if i % 4 == 0:
            assert got == o.find_all(host[i]), idx[i]
        total += len(got)
    assert total > 800


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["c3m16", "c3u"])
def test_round6_filter_forms_full_size(workload):
    """Round 6's filter forms at 10^7 rows -- c3m16: the WIDE filter (1000 Latin + 1000 Cyrillic + 1000 CJK keywords over mixed-script UTF-16
    rows); c3u: `(1000 keywords)[0-9]+`, find() of a pattern without bounded match lengths behind the filter (stride 4, the two-sided second
    level, starts by backward walks).  find() and containedIn() behind the filter against (1) the SAME calls with the filter switched off
    (needle_pattern_set_prefilter: the ordinary scan kernels, another code path) on ALL rows, (2) the oracle on rows sampled over the whole
    batch, (3) two unequal shards joined; the 2-byte and 4-byte result forms against the two arrays."""
    import torch
    from needle_amd.pattern import Pattern, unpack_bitmap
    p, rows, words = make(workload)
    n = rows.shape[0]
    fw, fs, fe = p.find_batch(rows)
    cw = p.contained_in_batch(rows)
    torch.cuda.synchronize()
    assert torch.equal(fw, cw)
    f_bits = unpack_bitmap(fw, n)
    assert 0.05 < f_bits.mean() < 0.6
    check_sample_against_oracle(p, rows, f_bits, fs.cpu().numpy(), fe.cpu().numpy(), unpack_bitmap(cw, n))
    partition_invariant(p.find_batch, rows, fw)
    w16, se16 = p.find_packed16_batch(rows)
    sev = se16.cpu().numpy().view(np.uint32)
    s_np, e_np = fs.cpu().numpy(), fe.cpu().numpy()
    assert torch.equal(w16, fw) and ((sev == 0xFFFFFFFF) == ~f_bits).all()
    assert ((sev & 0xFFFF)[f_bits] == s_np[f_bits]).all() and ((sev >> 16)[f_bits] == e_np[f_bits]).all()
    w8, se8 = p.find_packed8_batch(rows)
    s8, e8 = Pattern.unpack8(se8.cpu().numpy())
    assert torch.equal(w8, fw) and (s8 == s_np).all() and (e8 == e_np).all()
    # the same calls without the filter: every row
    p.set_prefilter(p.PREFILTER_OFF)
    try:
        gw, gs, ge = p.find_batch(rows)
        gc = p.contained_in_batch(rows)
        torch.cuda.synchronize()
    finally:
        p.set_prefilter(p.PREFILTER_AUTO)
    assert torch.equal(gw, fw) and torch.equal(gc, cw) and torch.equal(gs, fs) and torch.equal(ge, fe)
