"""The n-gram filter's FLOOD WATCH made observable and pinnable (include/needle_hip.h: needle_pattern_set_prefilter,
needle_pattern_prefilter_state).  Flood text -- rows built from the dictionary's own keyword TAILS behind a wrong first char: a
candidate at nearly every window -- must flip the state to "suspended" after the first evaluated launch; quiet text must bring the
filter back once the suspension has run out; ON never suspends, OFF never launches the filter kernel; and the answers are the
CPU oracle's (DFAClassBuilder.java:335-471, 625-659) in every state."""
import numpy as np
import pytest


@pytest.mark.gpu
def test_flood_flips_the_state_quiet_text_recovers_answers_never_change(oracle_lib):
    import torch
    from needle_amd import workload as W
    from needle_amd.pattern import DFACompiler, unpack_bitmap
    from test_compile_matches_txt import oracle_for
    words = W.keywords(1000, min_len=6, max_len=8)
    rx = "|".join(words)
    p = DFACompiler.compile(rx, "t", 0)
    o, _ = oracle_for(rx, 0)
    dev = "cuda"
    n, stride = 64 * 200 + 9, 256   # 3.2 MB of text per call: every call gives the watch more than the 1024 KiB it evaluates on
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    long_words = [w for w in words if len(w) >= 6][:512]
    wt8 = torch.zeros((len(long_words), 8), dtype=torch.uint8, device=dev) + 32
    for i, w in enumerate(long_words):
        t = torch.tensor([ord(c) for c in w[-8:]], dtype=torch.uint8, device=dev)
        t[0] = ord("q") if t[0] != ord("q") else ord("z")
        wt8[i, 8 - len(t):] = t
    flood = wt8[torch.randint(0, len(long_words), (n, stride // 8), device=dev, generator=g)].reshape(n, stride).clone()
    kw0 = torch.tensor([ord(c) for c in words[0]], dtype=torch.uint8, device=dev)
    flood[::5, 8:8 + len(kw0)] = kw0
    quiet = W.keyword_batch(torch, words, 9, n, stride, device=dev)

    def check(rows):
        fw, fs, fe = p.find_batch(rows)
        torch.cuda.synchronize()
        of, ofs, ofe = o.batch_find(rows.cpu().numpy(), threads=8)
        assert (unpack_bitmap(fw, n) == of).all() and (fs.cpu().numpy() == ofs).all() and (fe.cpu().numpy() == ofe).all()

    st = p.prefilter_state("forwards")
    assert st["mode"] == p.PREFILTER_AUTO and st["has_filter"] == 0 and st["filter_launches"] == 0
    check(flood)                                   # launch 1: the filter kernel; its counters arrive behind it
    st = p.prefilter_state("forwards")
    assert st["has_filter"] == 1 and st["filter_launches"] == 1 and st["suspended_calls_left"] == 0
    check(flood)                                   # call 2 evaluates launch 1: flooded -> this call and the next 31 take the scan kernel
    st = p.prefilter_state("forwards")
    assert st["last_candidates_per_kib"] > 16 and st["suspended_calls_left"] == 31 and st["backoff"] == 64 and st["filter_launches"] == 1, st
    for _ in range(31):                            # quiet text now: still the scan kernel while the suspension lasts
        p.find_batch(quiet)
    torch.cuda.synchronize()
    st = p.prefilter_state("forwards")
    assert st["suspended_calls_left"] == 0 and st["suspended_calls"] == 32 and st["filter_launches"] == 1, st
    check(quiet)                                   # the filter is tried again ...
    check(quiet)                                   # ... and found quiet: the backoff starts over
    st = p.prefilter_state("forwards")
    assert st["filter_launches"] == 3 and st["last_candidates_per_kib"] < 16 and st["backoff"] == 32 and st["suspended_calls_left"] == 0, st
    # pinned ON: flood text goes through the filter kernel every time (slow, never suspended) -- the answers do not move
    p.set_prefilter(p.PREFILTER_ON)
    for _ in range(3):
        check(flood)
    st = p.prefilter_state("forwards")
    assert st["mode"] == p.PREFILTER_ON and st["filter_launches"] == 6, st
    # pinned OFF: the ordinary kernels
    p.set_prefilter(p.PREFILTER_OFF)
    check(flood)
    check(quiet)
    cw = p.contained_in_batch(quiet)
    torch.cuda.synchronize()
    assert (unpack_bitmap(cw, n) == o.batch_contained_in(quiet.cpu().numpy(), threads=8)).all()
    st = p.prefilter_state("forwards")
    assert st["mode"] == p.PREFILTER_OFF and st["filter_launches"] == 6, st
    assert p.prefilter_state("contained_in")["filter_launches"] == 0
    p.set_prefilter(p.PREFILTER_AUTO)
    with pytest.raises(Exception):
        p.set_prefilter(7)
