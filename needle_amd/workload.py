"""Synthetic haystack batches of the BASELINE.json configs: a counter-based generator keyed by
(seed, row, column) so that any shard of any size is reproducible on CPU (numpy) and on the GPU (torch)
with bit-identical contents.  Integer-only; all arithmetic in int64 with explicit 32-bit masking."""
import numpy as np

SEED = 0x5EED1234
_M32 = 0xFFFFFFFF


def _hash32(x):
    """lowbias32 finaliser on int64 arrays/tensors holding values < 2**32 (works for numpy and torch)."""
    x = x & _M32
    x = ((x ^ (x >> 16)) * 0x7FEB352D) & _M32
    x = ((x ^ (x >> 15)) * 0x846CA68B) & _M32
    return x ^ (x >> 16)


def _h32i(x):
    """_hash32 on a Python int (same function, no numpy overflow warnings)."""
    x &= _M32
    x = ((x ^ (x >> 16)) * 0x7FEB352D) & _M32
    x = ((x ^ (x >> 15)) * 0x846CA68B) & _M32
    return x ^ (x >> 16)


def _grid(xp, row0, n_rows, n_cols, device=None):
    if xp is np:
        r = np.arange(row0, row0 + n_rows, dtype=np.int64)[:, None]
        c = np.arange(n_cols, dtype=np.int64)[None, :]
    else:
        r = xp.arange(row0, row0 + n_rows, dtype=xp.int64, device=device)[:, None]
        c = xp.arange(n_cols, dtype=xp.int64, device=device)[None, :]
    return r, c


def _lut(xp, values, device=None):
    a = np.asarray(values, dtype=np.int64)
    return a if xp is np else xp.tensor(a, dtype=xp.int64, device=device)


ALPHA_C2 = [ord(c) for c in "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz "]  # 53 symbols, no digits
ALPHA_C3 = [ord(c) for c in "abcdefghijklmnopqrstuvwxyz "]


def digits_batch(xp, row0, n_rows, n_cols=256, seed=SEED, device=None):
    """C2: chars uniform over [A-Za-z ]; in 50 % of rows (row-hash bit) a run of 1..6 digits at a uniform position."""
    r, c = _grid(xp, row0, n_rows, n_cols, device)
    base = _lut(xp, ALPHA_C2, device)[_hash32(seed + r * 1315423911 + c * 2654435761) % len(ALPHA_C2)]
    h = _hash32((seed ^ 0x9E3779B9) + r * 40503)
    plant = (h & 1) == 1
    run = 1 + ((h >> 1) % 6)
    pos = (h >> 8) % (n_cols - run + 1)
    digit = 48 + _hash32((seed ^ 0x85EBCA6B) + r * 69069 + c) % 10
    inside = plant & (c >= pos) & (c < pos + run)
    out = xp.where(inside, digit, base)
    return out.astype(np.uint8) if xp is np else out.to(xp.uint8)


def keywords(n=1000, seed=SEED, min_len=3, max_len=5):
    """C3: n distinct keywords of length min_len..max_len (3..5: SURVEY.md s8d) over [a-z], in generated order (order
    matters: leftmost-first).  The sparse-match variant of C3 uses 6..8: uniform [a-z ] text then matches a keyword
    by chance in < 0.1 % of the rows, so only the planted 25 % match and every lane stays live over the whole row."""
    out, seen, i = [], set(), 0
    while len(out) < n:
        h = _h32i((seed ^ 0xC3) + i * 7919)
        ln = min_len + h % (max_len - min_len + 1)
        w = "".join(chr(97 + _h32i(seed + i * 31 + j * 1000003) % 26) for j in range(ln))
        i += 1
        if w not in seen:
            seen.add(w)
            out.append(w)
    return out


def keyword_batch(xp, words, row0, n_rows, n_cols=256, seed=SEED, device=None):
    """C3: chars uniform over [a-z ]; one keyword planted in 25 % of rows."""
    r, c = _grid(xp, row0, n_rows, n_cols, device)
    base = _lut(xp, ALPHA_C3, device)[_hash32(seed + 77 + r * 1315423911 + c * 2654435761) % len(ALPHA_C3)]
    h = _hash32((seed ^ 0x51ED270B) + r * 40503)
    plant = (h & 3) == 0
    maxlen = max(len(w) for w in words)
    wtab = np.zeros((len(words), maxlen), dtype=np.int64)
    wlen = np.zeros(len(words), dtype=np.int64)
    for i, w in enumerate(words):
        wtab[i, :len(w)] = [ord(ch) for ch in w]
        wlen[i] = len(w)
    wtab, wlen = _lut(xp, wtab, device), _lut(xp, wlen, device)
    k = (h >> 2) % len(words)
    ln = wlen[k]
    pos = (h >> 12) % (n_cols - ln + 1)
    rel = c - pos
    inside = plant & (rel >= 0) & (rel < ln)
    relc = xp.clip(rel, 0, maxlen - 1) if xp is np else rel.clamp(0, maxlen - 1)
    kk = k + 0 * c  # broadcast to [n_rows, n_cols]
    planted = wtab[kk, relc]
    out = xp.where(inside, planted, base)
    return out.astype(np.uint8) if xp is np else out.to(xp.uint8)


# C5: mixed-script UTF-16.  Ranges the regex recognises (explicit BMP ranges; >= 40 ranges over several scripts)
SCRIPT_RANGES = [
    (0x0391, 0x03A1), (0x03A3, 0x03A9), (0x03B1, 0x03C1), (0x03C3, 0x03C9),  # Greek
    (0x0410, 0x041F), (0x0420, 0x042F), (0x0430, 0x043F), (0x0440, 0x044F), (0x0451, 0x0451), (0x0401, 0x0401),  # Cyrillic
    (0x05D0, 0x05DA), (0x05DB, 0x05EA),  # Hebrew
    (0x0531, 0x0556), (0x0561, 0x0586),  # Armenian
    (0x0905, 0x0914), (0x0915, 0x0939),  # Devanagari
    (0x0E01, 0x0E2E), (0x0E30, 0x0E3A),  # Thai
    (0x10D0, 0x10FA),  # Georgian
    (0x3041, 0x3096), (0x30A1, 0x30FA),  # Hiragana / Katakana
    (0x4E00, 0x4E3F), (0x4E80, 0x4EBF), (0x4F00, 0x4F3F), (0x5000, 0x503F), (0x5100, 0x513F), (0x5200, 0x523F),
    (0x5300, 0x533F), (0x5400, 0x543F), (0x5500, 0x553F), (0x5600, 0x563F), (0x5700, 0x573F), (0x5800, 0x583F),
    (0x5900, 0x593F), (0x5A00, 0x5A3F), (0x5B00, 0x5B3F), (0x5C00, 0x5C3F), (0x5D00, 0x5D3F), (0x5E00, 0x5E3F),
    (0xAC00, 0xAC7F), (0xAD00, 0xAD7F), (0xAE00, 0xAE7F),  # Hangul fragments
]


def script_regex(min_run=3):
    """C5 regex: a run of >= min_run chars from SCRIPT_RANGES, written with explicit ranges (no \\w / \\p)."""
    cls = "[" + "".join("%s-%s" % (chr(a), chr(b)) if a != b else chr(a) for a, b in SCRIPT_RANGES) + "]"
    return cls + "{%d}" % min_run + cls + "*"


def script_batch(xp, row0, n_rows, n_cols=256, seed=SEED, device=None):
    """C5: UTF-16 code units: 70 % ASCII letters/space, 15 % other BMP code points outside SCRIPT_RANGES (never
    U+FFFF), 15 % in-range chars; in 30 % of rows a run of 3..8 in-range chars is planted."""
    r, c = _grid(xp, row0, n_rows, n_cols, device)
    h = _hash32(seed + 5 + r * 1315423911 + c * 2654435761)
    sel = h % 100
    ascii_ch = _lut(xp, ALPHA_C2, device)[(h >> 8) % len(ALPHA_C2)]
    starts = _lut(xp, [a for a, _ in SCRIPT_RANGES], device)
    sizes = _lut(xp, [b - a + 1 for a, b in SCRIPT_RANGES], device)
    ri = (h >> 10) % len(SCRIPT_RANGES)
    in_range = starts[ri] + (h >> 17) % sizes[ri]
    other = 0x2000 + (h >> 9) % 0x0C00  # punctuation / symbols / box drawing blocks: outside every range above
    # isolated in-range chars are allowed only singly: keep them from forming runs by spacing (col parity)
    single = xp.where((c % 4) == 0, in_range, ascii_ch)
    base = xp.where(sel < 70, ascii_ch, xp.where(sel < 85, other, single))
    hr = _hash32((seed ^ 0xC5C5C5C5) + r * 40503)
    plant = (hr % 10) < 3
    run = 3 + (hr >> 4) % 6
    pos = (hr >> 8) % (n_cols - run + 1)
    inside = plant & (c >= pos) & (c < pos + run)
    out = xp.where(inside, in_range, base)
    return out.astype(np.uint16) if xp is np else out.to(xp.int16)


def url_strings(n=1000, seed=SEED):
    """C1: 50 % "http://"+1..40 chars of [a-z0-9./]; 25 % the same without the prefix; 25 % prefix + a newline inside."""
    alpha = "abcdefghijklmnopqrstuvwxyz0123456789./"
    out = []
    for i in range(n):
        h = _h32i((seed ^ 0xC1) + i * 7919)
        ln = 1 + h % 40
        body = "".join(alpha[_h32i(seed + i * 131 + j * 1000003) % len(alpha)] for j in range(ln))
        kind = (h >> 8) % 4
        if kind < 2:
            out.append("http://" + body)
        elif kind == 2:
            out.append(body)
        else:
            k = (h >> 12) % (ln + 1)
            out.append("http://" + body[:k] + "\n" + body[k:])
    return out
