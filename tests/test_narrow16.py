"""The UTF-16 -> Latin-1 narrowing of the filter kernel (needle_amd/csrc/needle_ngram.h narrow_pair / narrow_pair_patched / narrow16) restated
in numpy, instruction by instruction (v_perm_b32 selectors, the SWAR "high byte not zero" mask), against its definition: char c -> c if
c <= 0xFF else 0xFF.  The kernel's use of it is tested on the device (tests/test_gpu_prefilter_utf16.py); this pins the bit arithmetic."""
import numpy as np


def v_perm_b32(s0, s1, sel):
    """D.byte[i] = byte (sel.byte[i]) of the 8 bytes {s1 (0..3), s0 (4..7)} -- the selectors used here are all below 8."""
    pool = np.concatenate([s1.view(np.uint8).reshape(-1, 4), s0.view(np.uint8).reshape(-1, 4)], axis=1)
    idx = [(sel >> (8 * i)) & 0xFF for i in range(4)]
    return np.ascontiguousarray(pool[:, idx]).view(np.uint32).reshape(-1)


def narrow_pair(x, y):
    return v_perm_b32(y, x, 0x06040200)


def narrow_pair_patched(x, y, page=0, sub=0xFF):
    page4, sub4 = np.uint32(page * 0x01010101), np.uint32(sub * 0x01010101)
    lo, hi = v_perm_b32(y, x, 0x06040200), v_perm_b32(y, x, 0x07050301) ^ page4
    m = (((hi & np.uint32(0x7F7F7F7F)) + np.uint32(0x7F7F7F7F)) | hi) & np.uint32(0x80808080)
    m = m | (m - (m >> np.uint32(7)))
    return (lo & ~m) | (sub4 & m)


def test_narrowing_is_min_of_char_and_0xff():
    rng = np.random.default_rng(11)
    chars = rng.integers(0, 0x10000, size=(50000, 4), dtype=np.uint32)
    # every interesting neighbourhood: 0x00FF / 0x0100, a zero low byte under a high byte, 0x7Fxx / 0x80xx (the SWAR carry), 0xFFFF
    chars[:3000] = rng.choice(np.array([0x0000, 0x0061, 0x007F, 0x0080, 0x00FE, 0x00FF, 0x0100, 0x0161, 0x01FF, 0x7F00, 0x7FFF, 0x8000, 0x8061, 0xFF00, 0xFFFF],
                                       dtype=np.uint32), size=(3000, 4))
    chars[3000:8000] = rng.integers(0, 0x100, size=(5000, 4), dtype=np.uint32)  # all four chars Latin-1: the fast path's case
    x = (chars[:, 0] | chars[:, 1] << 16).astype(np.uint32)
    y = (chars[:, 2] | chars[:, 3] << 16).astype(np.uint32)
    want = np.minimum(chars, 0xFF).astype(np.uint32)
    want = want[:, 0] | want[:, 1] << 8 | want[:, 2] << 16 | want[:, 3] << 24
    assert (narrow_pair_patched(x, y) == want).all()
    latin = (chars <= 0xFF).all(axis=1)
    assert latin.sum() > 100 and (narrow_pair(x, y)[latin] == want[latin]).all()
    # the fast path's test: some high byte of the 8 dwords not zero <=> some char above 0xFF
    any_hi = ((x | y) & np.uint32(0xFF00FF00)) != 0
    assert (any_hi == ~latin).all()


def test_narrowing_to_another_page():
    """chars of page P -> their low byte, every other char -> the page's substitute (needle_api.cpp utf16_route)"""
    rng = np.random.default_rng(12)
    for page, sub in ((0x04, 0x2F), (0x05, 0x00), (0xFF, 0x80), (0x00, 0xFF)):
        chars = rng.integers(0, 0x10000, size=(20000, 4), dtype=np.uint32)
        chars[:8000] = (page << 8) | rng.integers(0, 0x100, size=(8000, 4), dtype=np.uint32)
        chars[8000:9000] = rng.choice(np.array([0x0000, 0x0020, 0x00FF, ((page ^ 1) << 8) | 0x30, ((page ^ 0x80) << 8) | 0x30, 0xFFFF, (page << 8) | sub], dtype=np.uint32), size=(1000, 4))
        x = (chars[:, 0] | chars[:, 1] << 16).astype(np.uint32)
        y = (chars[:, 2] | chars[:, 3] << 16).astype(np.uint32)
        want = np.where((chars >> 8) == page, chars & 0xFF, sub).astype(np.uint32)
        want = want[:, 0] | want[:, 1] << 8 | want[:, 2] << 16 | want[:, 3] << 24
        assert (narrow_pair_patched(x, y, page, sub) == want).all(), (page, sub)
