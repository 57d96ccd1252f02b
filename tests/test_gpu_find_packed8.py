"""needle_find_packed8_dev / _host (include/needle_hip.h): find() on rows of at most 256 chars with a row's start() / end()
(DFAClassBuilder.java:625-667) as ONE uint16 -- start | (end - start) << 8, 0xFFFF = no match, 0xFFFE = the match (0, 256) -- against the CPU
oracle and against needle_find_dev on the same rows, in every kernel that stores results: the tiled kernel (packed functions, pair table,
LDS tables), the register-resident short-row kernel, the n-gram filter kernel (8-bit rows, UTF-16 rows narrowed, the wide filter)."""
import numpy as np
import pytest

from test_gpu_configs import compiled


def _check(p, o, host, lens=None):
    import torch
    from needle_amd.pattern import Pattern, unpack_bitmap
    n = host.shape[0]
    rows = torch.from_numpy(host.view(np.int16) if host.dtype == np.uint16 else host).cuda()
    tl = None if lens is None else torch.from_numpy(lens.astype(np.int32)).cuda()
    w, sl = p.find_packed8_batch(rows, tl)
    w0, s0, e0 = p.find_batch(rows, tl)
    torch.cuda.synchronize()
    s, e = Pattern.unpack8(sl.cpu().numpy())
    assert (w.cpu().numpy() == w0.cpu().numpy()).all()
    assert (s == s0.cpu().numpy()).all() and (e == e0.cpu().numpy()).all()
    m, os_, oe = o.batch_find(host, lens, threads=8)
    assert (unpack_bitmap(w, n) == m).all() and (s == os_).all() and (e == oe).all()
    if n <= 70000:  # the host entry point (upload -> kernel -> 2 bytes per row back)
        wh, slh = p.find_packed8_host(host, lens)
        sh, eh = Pattern.unpack8(slh)
        assert (unpack_bitmap(wh, n) == m).all() and (sh == os_).all() and (eh == oe).all()
    return int(m.sum()), s, e


@pytest.mark.gpu
@pytest.mark.parametrize("regex,width", [("[0-9]+", 256), ("[0-9]+", 48), ("[0-9]+", 16), ("(ab|cd)+e?", 64), ("[A-Za-z0-9 ]+", 256), ("x*", 256),
                                         ("Sherlock|Holmes|Watson|Irene|Adler|John|Baker", 128)])
@pytest.mark.parametrize("ragged", [False, True])
def test_packed8_find_equals_find_and_oracle(regex, width, ragged):
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
        if regex == "[A-Za-z0-9 ]+":
            host[1::4, width // 2] = ord("!")  # (the other rows match from 0 to their end: (0, 256) on full 256-char rows -- the escape)
        lens = ((np.arange(n, dtype=np.uint64) * 2654435761) % (width + 1)).astype(np.uint32) if ragged else None
        k, s, e = _check(p, o, host, lens)
        assert n < 1000 or k > n // 20
        if regex == "[A-Za-z0-9 ]+" and width == 256 and not ragged and n > 1000:
            assert ((s == 0) & (e == 256)).sum() > n // 2
        if regex == "x*" and n > 1000:
            assert (e == s).sum() > n // 2  # empty matches: length 0


@pytest.mark.gpu
def test_packed8_behind_the_filters_and_limits():
    """The 1000-keyword dictionary behind the n-gram filter on 8-bit rows, on UTF-16 rows (narrowed), a mixed-script dictionary behind the wide
    filter; rows beyond 256 chars are refused."""
    import torch
    from needle_amd import workload as W
    from needle_amd.pattern import DFACompiler, PatternException
    from test_compile_matches_txt import oracle_for
    words = W.keywords(1000, min_len=6, max_len=8)
    rx = "|".join(words)
    p = DFACompiler.compile(rx, "Keywords1k")
    o, _ = oracle_for(rx, 0)
    for n, ragged in ((64 * 400 + 9, False), (64 * 300 + 1, True)):
        host = W.keyword_batch(np, words, 5, n, 256)
        host[::9, 256 - len(words[1]):] = [ord(c) for c in words[1]]
        lens = ((np.arange(n, dtype=np.uint64) * 2654435761) % 257).astype(np.uint32) if ragged else None
        before = p.prefilter_state("forwards")["filter_launches"]
        assert _check(p, o, host, lens)[0] > n // 10
        assert p.prefilter_state("forwards")["filter_launches"] > before
        assert _check(p, o, host.astype(np.uint16), lens)[0] > n // 10
    mixed = W.keywords_mixed(200)
    rx = "|".join(mixed)
    p = DFACompiler.compile(rx, "Mixed")
    o, _ = oracle_for(rx, 0)
    host = W.mixed_keyword_batch(np, mixed, 9, 64 * 200 + 5, 256)
    host[::9, 256 - len(mixed[2]):] = [ord(c) for c in mixed[2]]
    assert _check(p, o, host, None)[0] > 2000
    rows = torch.zeros((64, 272), dtype=torch.uint8, device="cuda")
    with pytest.raises(PatternException):
        p.find_packed8_batch(rows)
