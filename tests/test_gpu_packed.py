"""Haystacks packed back to back (one char buffer + offsets): the device-side conversion to the
fixed-stride layout and the packed host entry points, against the oracle on the same strings.  Covers empty rows,
rows of every length mod 16, unaligned row starts, 8- and 16-bit code units, a row longer than the chosen stride."""
import numpy as np
import pytest

from test_gpu_configs import compiled


def random_rows(rng, n, max_len, alphabet):
    lens = rng.integers(0, max_len + 1, n)
    lens[:40] = np.arange(40) % (max_len + 1)  # every small length, incl. 0
    rows = [rng.choice(alphabet, int(l)) for l in lens]
    return rows, lens


def pack(rows, dtype):
    offsets = np.zeros(len(rows) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in rows])
    data = np.concatenate(rows).astype(dtype) if len(rows) and offsets[-1] else np.zeros(0, dtype=dtype)
    return data, offsets


def padded(rows, dtype):
    m = max(1, max(len(r) for r in rows))
    out = np.zeros((len(rows), m), dtype=dtype)
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("cw,regex,alphabet", [
    (1, "[0-9]+", [ord(c) for c in "abcxyz 0123456789"]),
    (1, "Sherlock|Holmes|Watson|Irene|Adler|John|Baker", [ord(c) for c in "SherlockHmsWatnIdJB "]),
    (2, "[α-ω]{2}[α-ω]*", [0x61, 0x62, 0x20, 0x3B1, 0x3B2, 0x3C9, 0x4E00, 0xFFFF]),
])
def test_packed_rows_equal_oracle(cw, regex, alphabet):
    import torch
    from needle_amd.pattern import unpack_bitmap
    p, o = compiled(regex)
    rng = np.random.default_rng(11 + cw)
    dtype = np.uint8 if cw == 1 else np.uint16
    rows, lens = random_rows(rng, 5000, 90, np.array(alphabet))
    data, offsets = pack(rows, dtype)
    ref = padded(rows, dtype)
    L = lens.astype(np.uint32)
    want_m = o.batch_matches(ref, L, threads=4)
    want_c = o.batch_contained_in(ref, L, threads=4)
    want_f, want_s, want_e = o.batch_find(ref, L, threads=4)
    # 1. packed host entry points
    n = len(rows)
    assert (unpack_bitmap(p.matches_packed(data, offsets), n) == want_m).all()
    assert (unpack_bitmap(p.contained_in_packed(data, offsets), n) == want_c).all()
    fw, fs, fe = p.find_packed(data, offsets)
    assert (unpack_bitmap(fw, n) == want_f).all() and (fs == want_s).all() and (fe == want_e).all()
    # 2. device-side conversion: exact layout (zero padded) + lengths, then the ordinary device entry points
    td = torch.from_numpy(data.view(np.int16) if cw == 2 else data).cuda()
    to = torch.from_numpy(offsets.astype(np.int64)).cuda()
    drows, dlen, ovf = p.rows_from_packed(td, to)
    assert int(ovf.item()) == 0
    assert (dlen.cpu().numpy() == lens).all()
    got = drows.cpu().numpy().view(dtype)
    assert got.shape[1] % (16 // cw) == 0
    assert (got[:, :ref.shape[1]] == ref).all() and not got[:, ref.shape[1]:].any()
    w2, s2, e2 = p.find_batch(drows, dlen)
    assert (unpack_bitmap(w2, n) == want_f).all() and (s2.cpu().numpy() == want_s).all() and (e2.cpu().numpy() == want_e).all()
    # 3. a stride shorter than the longest row: truncation is flagged, lengths are clamped
    short = 16 // cw * 2
    _r, l3, ovf3 = p.rows_from_packed(td, to, row_stride=short)
    assert int(ovf3.item()) == 1 and (l3.cpu().numpy() == np.minimum(lens, short)).all()


@pytest.mark.gpu
def test_find_strings_matches_single_string_matcher():
    from needle_amd.pattern import DFACompiler
    p = DFACompiler.compile("http://.+")
    strings = ["http://www.google.com", "", "nothing here", "see http://Γειά σου.com now", "http://", "xhttp://a\nhttp://b"]
    matched, st, en = p.find_strings(strings)
    for i, s in enumerate(strings):
        m = p.matcher(s)
        assert bool(matched[i]) == m.find(), s
        if matched[i]:
            assert (st[i], en[i]) == (m.start(), m.end()), s


@pytest.mark.gpu
def test_skewed_lengths_run_as_length_classes():
    """One 300 000-char document among thousands of short strings: padding every row to the longest one would need
    1.5 GB for 0.5 MB of text; the packed host path groups rows into length classes instead.  Same bits either way."""
    from needle_amd.pattern import unpack_bitmap
    p, o = compiled("[0-9]+")
    rng = np.random.default_rng(3)
    alphabet = np.array([ord(c) for c in "abcxyz 0123456789"])
    rows, lens = random_rows(rng, 5000, 60, alphabet)
    for at, n_chars in ((17, 300_000), (4000, 5_000), (4999, 700)):
        rows[at] = rng.choice(alphabet, n_chars)
    rows[17][:299_990] = ord("q")  # the long row's only digits sit in its last ten chars
    data, offsets = pack(rows, np.uint8)
    fw, fs, fe = p.find_packed(data, offsets)
    cb = unpack_bitmap(p.contained_in_packed(data, offsets), len(rows))
    mb = unpack_bitmap(p.matches_packed(data, offsets), len(rows))
    got = unpack_bitmap(fw, len(rows))
    for i, r in enumerate(rows):
        s = r.astype(np.uint8).tobytes().decode("latin-1")
        found, st, en = o.find(s)
        assert got[i] == found and cb[i] == found and mb[i] == o.matches(s), i
        if found:
            assert (fs[i], fe[i]) == (st, en), i
        else:
            assert (fs[i], fe[i]) == (-1, -1), i
    assert fs[17] >= 299_990


@pytest.mark.gpu
def test_large_host_batches_are_chunked():
    """Host batches larger than the device-resident chunk size run chunk by chunk (64-row aligned); the chunk size is read
    once per process, so the small-chunk run happens in a subprocess."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, ".")
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler
p = DFACompiler.compile("[0-9]+", "d")
rows = W.digits_batch(np, 0, 30011, 256)
lens = (np.arange(30011) * 2654435761 % 257).astype(np.uint32)
c = p.contained_in_batch(rows, lens)
m = p.matches_batch(rows)
fw, fs, fe = p.find_batch(rows, lens)
cnt, als, ale, more = p.find_all_dense(rows, 3, lens)  # needle_find_all_host, chunked the same way
co, cs, ce = p.find_all_csr(rows, lens)                 # needle_find_all_csr_host: offsets run on across the chunks
assert co[-1] == len(cs) == len(ce) and (np.diff(co)[:100] >= cnt[:100]).all()
np.save(sys.argv[1], np.concatenate([c.view(np.int64), m.view(np.int64), fw.view(np.int64), fs.astype(np.int64), fe.astype(np.int64),
                                     cnt.astype(np.int64), als.astype(np.int64).ravel(), ale.astype(np.int64).ravel(), np.array([int(more)]),
                                     co.astype(np.int64), cs.astype(np.int64), ce.astype(np.int64)]))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for chunk in ("0", str(1 << 20)):  # 0: library default (one chunk), 1 MiB: 8 chunks of 4096 rows
        path = "/tmp/needle_chunk_%s.npy" % chunk
        env = dict(os.environ)
        if chunk != "0":
            env["NEEDLE_HOST_CHUNK_BYTES"] = chunk
        subprocess.check_call([sys.executable, "-c", code, path], cwd=root, env=env)
        out.append(np.load(path))
    assert (out[0] == out[1]).all()


@pytest.mark.gpu
def test_one_pattern_shared_by_host_threads():
    """A Pattern is immutable and shareable (Pattern thread-safe, Matcher not): several host threads
    drive their own Matchers and small batches through ONE pattern object concurrently (ctypes drops the GIL)."""
    import threading
    from needle_amd.pattern import DFACompiler, unpack_bitmap
    p = DFACompiler.compile("[0-9]+|Sherlock")
    strings = ["ab12cd345", "", "no digits here", "Sherlock Holmes 221B", "x" * 100 + "7", "Sherloc"]
    want = [[(2, 4), (6, 9)], [], [], [(0, 8), (16, 19)], [(100, 101)], []]
    errors = []

    def worker(seed):
        try:
            for it in range(150):
                i = (seed + it) % len(strings)
                m = p.matcher(strings[i])
                got = []
                while m.find():
                    got.append((m.start(), m.end()))
                assert got == want[i], (seed, it, got)
                matched, st, en = p.find_strings(strings)
                assert list(matched) == [bool(w) for w in want]
                assert [(int(a), int(b)) for a, b, ok in zip(st, en, matched) if ok] == [w[0] for w in want if w]
        except Exception as e:  # noqa: BLE001 - reported below
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
