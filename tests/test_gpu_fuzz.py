"""Differential fuzz on the GPU: seeded random regexes (the generator of tests/test_compile_vs_python_restatement.py)
x random short haystacks over a small alphabet, all three ops through the packed entry points (ragged rows: the
guarded kernels) and through a fixed-stride full-length batch (the unguarded kernels), against the oracle walking the
same tables.  Whatever device mode the automaton lowers to (packed functions, pair table, uint8 / uint16 table) must
give the reference's bits."""
import random

import numpy as np
import pytest

from test_compile_vs_python_restatement import FLAG_SETS, random_regex
from test_gpu_configs import compiled

ALPHABET = [ord(c) for c in "abcxyz019 AB_\n."] + [0xE9, 0x416, 0x4E2D, 0xFFFF]


@pytest.mark.gpu
# (5009, 5038, 5054: seeds on which round 3's find-all "lengths" form first disagreed with the reference -- order-sensitive
# alternations, where indexBackwards does not report the longest match)
@pytest.mark.parametrize("seed", list(range(4)) + [5009, 5038, 5054])
def test_random_regexes_on_random_haystacks(seed):
    import torch
    from needle_amd.pattern import PatternException, unpack_bitmap
    rng = random.Random(4000 + seed)
    nrng = np.random.default_rng(seed)
    modes, done = set(), 0
    while done < 12:
        regex, flags = random_regex(rng), rng.choice(FLAG_SETS)
        try:
            p, o = compiled(regex, flags)
        except (PatternException, ValueError):
            continue
        done += 1
        modes.add(p.info()["kernel_mode"]["forwards"])
        # ragged, packed UTF-16 rows
        n = 400
        lens = nrng.integers(0, 70, n)
        rows = [nrng.choice(ALPHABET, int(l)).astype(np.uint16) for l in lens]
        offsets = np.zeros(n + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum(lens)
        data = np.concatenate(rows) if offsets[-1] else np.zeros(0, dtype=np.uint16)
        pad = np.zeros((n, 70), dtype=np.uint16)
        for i, r in enumerate(rows):
            pad[i, :len(r)] = r
        L = lens.astype(np.uint32)
        assert (unpack_bitmap(p.matches_packed(data, offsets), n) == o.batch_matches(pad, L, threads=4)).all(), (regex, flags)
        assert (unpack_bitmap(p.contained_in_packed(data, offsets), n) == o.batch_contained_in(pad, L, threads=4)).all(), (regex, flags)
        fw, fs, fe = p.find_packed(data, offsets)
        of, os_, oe = o.batch_find(pad, L, threads=4)
        assert (unpack_bitmap(fw, n) == of).all() and (fs == os_).all() and (fe == oe).all(), (regex, flags)
        # full-length 8-bit rows, stride 128: the unguarded kernels
        full = nrng.choice([c for c in ALPHABET if c < 256], (1024, 128)).astype(np.uint8)
        t = torch.from_numpy(full).cuda()
        assert (unpack_bitmap(p.matches_batch(t), 1024) == o.batch_matches(full, threads=4)).all(), (regex, flags)
        assert (unpack_bitmap(p.contained_in_batch(t), 1024) == o.batch_contained_in(full, threads=4)).all(), (regex, flags)
        fw, fs, fe = p.find_batch(t)
        of, os_, oe = o.batch_find(full, threads=4)
        assert (unpack_bitmap(fw, 1024) == of).all(), (regex, flags)
        assert (fs.cpu().numpy() == os_).all() and (fe.cpu().numpy() == oe).all(), (regex, flags)
        # every non-overlapping match of every row (the one-pass find-all kernel, compact form), full and ragged UTF-16 rows
        pad72 = np.zeros((n, 72), dtype=np.uint16)  # device batches: row stride a multiple of 16 bytes
        pad72[:, :70] = pad
        for rows_t, lens_t, host, host_lens in ((t, None, full, None),
                                                (torch.from_numpy(pad72.view(np.int16)).cuda(), torch.from_numpy(L.astype(np.int32)).cuda(), pad72, L)):
            sub = range(0, host.shape[0], 3)
            offs, s_all, e_all = p.find_all_batch(rows_t, lens_t)
            offs, s_all, e_all = offs.cpu().numpy(), s_all.cpu().numpy(), e_all.cpu().numpy()
            for i in sub:
                want = o.find_all(host[i] if host_lens is None else host[i, :host_lens[i]])
                got = list(zip(s_all[offs[i]:offs[i + 1]].tolist(), e_all[offs[i]:offs[i + 1]].tolist()))
                assert got == want, (regex, flags, "find-all", i, got[:4], want[:4])
        # short rows (stride <= 64 B: the register-resident kernel), ragged and full
        lens_s = nrng.integers(0, 25, n)
        rows_s = [nrng.choice(ALPHABET, int(l)).astype(np.uint16) for l in lens_s]
        off_s = np.zeros(n + 1, dtype=np.uint64)
        off_s[1:] = np.cumsum(lens_s)
        data_s = np.concatenate(rows_s) if off_s[-1] else np.zeros(0, dtype=np.uint16)
        pad_s = np.zeros((n, 25), dtype=np.uint16)
        for i, r in enumerate(rows_s):
            pad_s[i, :len(r)] = r
        fw, fs, fe = p.find_packed(data_s, off_s)
        of, os_, oe = o.batch_find(pad_s, lens_s.astype(np.uint32), threads=4)
        assert (unpack_bitmap(fw, n) == of).all() and (fs == os_).all() and (fe == oe).all(), (regex, flags, "short")
        assert (unpack_bitmap(p.matches_packed(data_s, off_s), n) == o.batch_matches(pad_s, lens_s.astype(np.uint32), threads=4)).all(), (regex, flags, "short")
        full_s = nrng.choice([c for c in ALPHABET if c < 256], (1000, 48)).astype(np.uint8)
        ts = torch.from_numpy(full_s).cuda()
        fw, fs, fe = p.find_batch(ts)
        of, os_, oe = o.batch_find(full_s, threads=4)
        assert (unpack_bitmap(fw, 1000) == of).all() and (fs.cpu().numpy() == os_).all() and (fe.cpu().numpy() == oe).all(), (regex, flags, "short full")
        assert (unpack_bitmap(p.contained_in_batch(ts), 1000) == o.batch_contained_in(full_s, threads=4)).all(), (regex, flags, "short full")
        # few long rows (the stripe path when the automaton lowers to packed functions; one row per lane otherwise)
        # (67 stripes of 4 KiB: long enough for the speculative-stripe path of the table-mode automata as well)
        long_rows = nrng.choice([c for c in ALPHABET if c < 256], (3, 274_432)).astype(np.uint8)
        long_lens = np.array([274_432, 200_001, 4096], dtype=np.uint32)
        tl = torch.from_numpy(long_rows).cuda()
        tll = torch.from_numpy(long_lens.astype(np.int32)).cuda()
        fw, fs, fe = p.find_batch(tl, tll)
        of, os_, oe = o.batch_find(long_rows, long_lens, threads=3)
        assert (unpack_bitmap(fw, 3) == of).all() and (fs.cpu().numpy() == os_).all() and (fe.cpu().numpy() == oe).all(), (regex, flags, "long")
        assert (unpack_bitmap(p.matches_batch(tl, tll), 3) == o.batch_matches(long_rows, long_lens, threads=3)).all(), (regex, flags, "long")
        assert (unpack_bitmap(p.contained_in_batch(tl, tll), 3) == o.batch_contained_in(long_rows, long_lens, threads=3)).all(), (regex, flags, "long")
    assert modes  # (which device modes a seed draws varies: packed functions, pair table, uint8 table)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,lds", [(202, "20000"), (205, "20000"), (210, "12000"), (233, "40000"), (235, "20000")])
def test_random_dictionaries(seed, lds):
    """scripts/dictionary_fuzz.py: random keyword unions whose keywords are prefixes / suffixes / extensions of one another, planted
    anywhere, at the very end of a row and cut by it, full and ragged rows, second find() from the cursor -- against the oracle.
    The seeds here land in the compressed automaton (mode 6) with its lengths program (END records, D_L rows as default rows)
    and, the last one, without it."""
    import importlib.util
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("dictionary_fuzz", os.path.join(root, "scripts", "dictionary_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    env = dict(os.environ)
    if lds:
        env["NEEDLE_MAX_PROG_LDS"] = lds
    r = subprocess.run([sys.executable, "-c", mod.CODE, str(seed)], env=env, capture_output=True, text=True, cwd=root, timeout=900)
    assert "DICT-OK" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
    assert ("mode 6, lengths form 1" if seed != 235 else "mode 6, lengths form 0") in r.stdout, r.stdout[-300:]  # (235: it does not fit 20 KB)
