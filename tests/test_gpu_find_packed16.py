"""needle_find_packed16_dev (include/needle_hip.h): find() whose scan kernel stores a row's start() / end()
(DFAClassBuilder.java:625-667) as ONE dword, start | end << 16, 0xFFFFFFFF = no match -- against the CPU oracle and against
needle_find_dev on the same rows, in every kernel that stores results: the tiled kernel (packed functions, pair table, LDS
tables, the compressed automaton), the register-resident short-row kernel, the n-gram filter kernel, and -- through scratch +
one pack pass -- the stripe paths for few long rows."""
import numpy as np
import pytest

from test_gpu_configs import compiled


def _unpack(se):
    se = se.view(np.uint32)
    lo, hi = (se & 0xFFFF).astype(np.int64), (se >> 16).astype(np.int64)
    return np.where(lo == 0xFFFF, -1, lo), np.where(hi == 0xFFFF, -1, hi)


def _check(p, o, host, lens=None):
    import torch
    from needle_amd.pattern import unpack_bitmap
    n = host.shape[0]
    rows = torch.from_numpy(host).cuda()
    tl = None if lens is None else torch.from_numpy(lens.astype(np.int32)).cuda()
    w, se = p.find_packed16_batch(rows, tl)
    w0, s0, e0 = p.find_batch(rows, tl)
    torch.cuda.synchronize()
    s, e = _unpack(se.cpu().numpy())
    assert (w.cpu().numpy() == w0.cpu().numpy()).all()
    assert (s == s0.cpu().numpy()).all() and (e == e0.cpu().numpy()).all()
    m, os_, oe = o.batch_find(host, lens, threads=8)
    assert (unpack_bitmap(w, n) == m).all() and (s == os_).all() and (e == oe).all()
    return int(m.sum())


@pytest.mark.gpu
@pytest.mark.parametrize("regex,width", [("[0-9]+", 256), ("[0-9]+", 48), ("[0-9]+", 16), ("(ab|cd)+e?", 64),
                                         ("Sherlock|Holmes|Watson|Irene|Adler|John|Baker", 128)])
@pytest.mark.parametrize("ragged", [False, True])
def test_packed16_find_equals_find_and_oracle(regex, width, ragged):
    from needle_amd import workload as W
    p, o = compiled(regex)
    for n in (1, 63, 64 * 700 + 13, 200_000):
        host = W.digits_batch(np, 23, n, width).copy()
        if "Sherlock" in regex:
            host[::5, 3:11] = np.frombuffer(b"Sherlock", dtype=np.uint8)
            host[2::7, width - 6:width] = np.frombuffer(b"Watson", dtype=np.uint8)
        if "ab" in regex:
            host[::3, 5:9] = np.frombuffer(b"abcd", dtype=np.uint8)
            host[1::11, width - 3:width] = np.frombuffer(b"cde", dtype=np.uint8)
        lens = ((np.arange(n, dtype=np.uint64) * 2654435761) % (width + 1)).astype(np.uint32) if ragged else None
        k = _check(p, o, host, lens)
        assert n < 1000 or k > n // 20


@pytest.mark.gpu
@pytest.mark.parametrize("min_len,max_len", [(3, 8), (6, 8)])
def test_packed16_find_big_dictionary(min_len, max_len):
    """The 1000-keyword union: LDS table u16 (3..8 chars) / the compressed automaton behind the n-gram filter (6..8 chars)."""
    import torch
    from needle_amd import workload as W
    from needle_amd.pattern import DFACompiler
    from oracle.walker import Dfa, OraclePattern
    words = W.keywords(1000, min_len=min_len, max_len=max_len)
    p = DFACompiler.compile("|".join(words), "Keywords1k")
    t = p.tables()
    d = {k: Dfa(t["class_map"], t["stride"], v["table"], v["accepting"], v["max_char"]) for k, v in t["dfas"].items()}
    o = OraclePattern(d["matches"], d["contained_in"], d["forwards"], d["backwards"], t["fixed_len"], -1)
    for n, ragged in ((64 * 400 + 9, False), (64 * 300 + 1, True)):
        host = W.keyword_batch(torch, words, 5, n, 256, device="cuda").cpu().numpy()
        lens = ((np.arange(n, dtype=np.uint64) * 2654435761) % 257).astype(np.uint32) if ragged else None
        assert _check(p, o, host, lens) > n // 10


@pytest.mark.gpu
def test_packed16_find_long_rows_and_limits():
    """Few long rows take the stripe paths (int32 results in scratch + one pack pass); rows beyond 65 534 chars are refused."""
    import torch
    from needle_amd import workload as W
    for regex in ("[0-9]+", "Sherlock|Holmes|Watson|Irene|Adler|John|Baker"):
        p, o = compiled(regex)
        host = W.digits_batch(np, 9, 96, 65520).copy()
        host[::4, 40000:40008] = np.frombuffer(b"Sherlock", dtype=np.uint8)
        _check(p, o, host)
        lens = (65520 - (np.arange(96, dtype=np.uint32) * 977) % 60000).astype(np.uint32)
        _check(p, o, host, lens)
    p, _ = compiled("[0-9]+")
    rows = torch.zeros((64, 65600), dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception):
        p.find_packed16_batch(rows)
    # the limit is on the rows, not on the stride they are stored with (host strides are padded to 16 bytes)
    host = W.digits_batch(np, 4, 70, 65530)
    w, se = p.find_packed16_host(host)
    _, s0, e0 = p.find_batch(host)
    s, e = _unpack(se)
    assert (s == s0).all() and (e == e0).all()
    hw, rec = p.find_compact(host)
    assert (hw == w).all() and (rec["start"] == s0[s0 >= 0]).all() and (rec["end"] == e0[s0 >= 0]).all()
