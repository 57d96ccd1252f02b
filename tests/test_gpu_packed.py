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
