"""Every distinct pattern of the reference's matches.txt (tests/golden/matches.json: leftmost-first / leftmost-longest
alternations, counted repetitions, char sets, unicode, case-insensitivity ...) over a seeded batch of haystacks
built from the golden haystacks themselves (the originals, their prefixes / suffixes / concatenations, and
shuffles): all three ops as one packed GPU batch against the oracle walking the same tables."""
import random

import numpy as np
import pytest

from test_compile_matches_txt import DOC, flag_sets
from test_gpu_configs import compiled


def haystack_pool(rng):
    base = [r["haystack"] for r in DOC["rows"]]
    out = list(base)
    for _ in range(400):
        a, b = rng.choice(base), rng.choice(base)
        k = rng.random()
        if k < 0.3:
            out.append(a + b)
        elif k < 0.5:
            out.append(a[:rng.randint(0, len(a))])
        elif k < 0.7:
            out.append(a[rng.randint(0, len(a)):] + b[:rng.randint(0, len(b))])
        else:
            chars = list(a + b)
            rng.shuffle(chars)
            out.append("".join(chars))
    return out


@pytest.mark.gpu
def test_every_matches_txt_pattern_over_a_batch():
    from needle_amd.pattern import PatternException, pack_strings, unpack_bitmap
    rng = random.Random(99)
    pool = haystack_pool(rng)
    data, offsets = pack_strings(pool)
    n = len(pool)
    width = max(len(h) for h in pool)
    pad = np.zeros((n, max(1, width)), dtype=np.uint16)
    lens = np.zeros(n, dtype=np.uint32)
    for i, h in enumerate(pool):
        u = np.frombuffer(h.encode("utf-16-le", "surrogatepass"), dtype=np.uint16)
        pad[i, :u.size] = u
        lens[i] = u.size
    seen, done = set(), 0
    for row in DOC["rows"]:
        for flags in flag_sets(row):
            key = (row["pattern"], flags)
            if key in seen:
                continue
            seen.add(key)
            try:
                p, o = compiled(row["pattern"], flags)
            except PatternException:
                continue
            done += 1
            assert (unpack_bitmap(p.matches_packed(data, offsets), n) == o.batch_matches(pad, lens, threads=4)).all(), key
            assert (unpack_bitmap(p.contained_in_packed(data, offsets), n) == o.batch_contained_in(pad, lens, threads=4)).all(), key
            fw, fs, fe = p.find_packed(data, offsets)
            of, os_, oe = o.batch_find(pad, lens, threads=4)
            assert (unpack_bitmap(fw, n) == of).all() and (fs == os_).all() and (fe == oe).all(), key
    assert done > 150
