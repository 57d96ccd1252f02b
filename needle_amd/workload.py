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


# c3m16: a keyword dictionary over THREE scripts on many pages of the BMP -- Latin (page 0), Cyrillic (page 4), CJK ideographs (20 of them,
# one per page 0x4E .. 0x9B) -- for UTF-16 rows.  The reference's class map covers all 65 536 code units of any pattern (DFA.java:438-463);
# 26 + 32 + 20 letters + the gaps between their ranges stay below its 127 usable classes (ByteClassUtil.java:126-128: signed bytes).
ALPHA_LATIN = [ord(c) for c in "abcdefghijklmnopqrstuvwxyz"]
ALPHA_CYRILLIC = list(range(0x0430, 0x0450))
ALPHA_CJK = [0x4E00 + 0x3FD * k for k in range(20)]
SCRIPTS = [ALPHA_LATIN, ALPHA_CYRILLIC, ALPHA_CJK]


def keywords_mixed(n_per_script=1000, seed=SEED, min_len=6, max_len=8):
    """n_per_script distinct keywords of min_len..max_len code units per script, scripts interleaved (latin, cyrillic, cjk, latin, ...)."""
    per = []
    for si, alpha in enumerate(SCRIPTS):
        out, seen, i = [], set(), 0
        while len(out) < n_per_script:
            h = _h32i((seed ^ (0xC3 + 977 * si)) + i * 7919)
            ln = min_len + h % (max_len - min_len + 1)
            w = "".join(chr(alpha[_h32i(seed + 13 * si + i * 31 + j * 1000003) % len(alpha)]) for j in range(ln))
            i += 1
            if w not in seen:
                seen.add(w)
                out.append(w)
        per.append(out)
    return [per[k % 3][k // 3] for k in range(3 * n_per_script)]


def mixed_keyword_batch(xp, words, row0, n_rows, n_cols=256, seed=SEED, device=None):
    """c3m16: UTF-16 rows; a row's script is hashed from its number, 85 % of its chars come from that script's letters + space, 15 % from
    the other two (mixed-script text); one keyword (of any script) planted in 25 % of the rows."""
    r, c = _grid(xp, row0, n_rows, n_cols, device)
    h = _hash32(seed + 177 + r * 1315423911 + c * 2654435761)
    hr = _hash32((seed ^ 0x3C3C16) + r * 40503)
    alphas = [a + [32] for a in SCRIPTS]
    size = max(len(a) for a in alphas)
    tab = np.zeros((3, size), dtype=np.int64)
    lens = np.zeros(3, dtype=np.int64)
    for i, a in enumerate(alphas):
        tab[i, :len(a)] = a
        lens[i] = len(a)
    tab, lens = _lut(xp, tab, device), _lut(xp, lens, device)
    script = (hr >> 3) % 3 + 0 * c
    stray = (h % 100) >= 85
    script = xp.where(stray, (script + 1 + (h >> 7) % 2) % 3, script)
    base = tab[script, (h >> 9) % lens[script]]
    plant = (hr & 3) == 0
    maxlen = max(len(w) for w in words)
    wtab = np.zeros((len(words), maxlen), dtype=np.int64)
    wlen = np.zeros(len(words), dtype=np.int64)
    for i, w in enumerate(words):
        wtab[i, :len(w)] = [ord(ch) for ch in w]
        wlen[i] = len(w)
    wtab, wlen = _lut(xp, wtab, device), _lut(xp, wlen, device)
    k = (hr >> 5) % len(words)
    ln = wlen[k]
    pos = (hr >> 15) % (n_cols - ln + 1)
    rel = c - pos
    inside = plant & (rel >= 0) & (rel < ln)
    relc = xp.clip(rel, 0, maxlen - 1) if xp is np else rel.clamp(0, maxlen - 1)
    planted = wtab[k + 0 * c, relc]
    out = xp.where(inside, planted, base)
    return out.astype(np.uint16) if xp is np else out.to(xp.int16)


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


# C5w ("wide"): what C5 was meant to stress (tens of classes, tens of states) -- per-script runs IN SEQUENCE, so
# that every script (and several sub-ranges of it) is a char class of its own and the automaton has to remember where in which
# alternative it is: UTF-16 rows go ptab -> page -> column -> LDS table.  Explicit ranges only (no \\w / \\p; the reference's parser has {n} and {n,m}, no {n,}).
SEQ_ALTS = [
    # (regex alternative, [(lo, hi, min, max) per element of a planted instance])
    ("[\u0391-\u03a1][\u03b1-\u03c1\u03c3-\u03c9]{2}[\u03b1-\u03c1\u03c3-\u03c9]*", [(0x0391, 0x03A1, 1, 1), (0x03B1, 0x03C1, 2, 5)]),              # Greek capital, then lower
    ("[\u0410-\u042f][\u0430-\u044f]+\u0451?", [(0x0410, 0x042F, 1, 1), (0x0430, 0x044F, 1, 5)]),                        # Cyrillic
    ("[\u05d0-\u05ea]{3}[\u05d0-\u05ea]*", [(0x05D0, 0x05EA, 3, 6)]),                                                                      # Hebrew
    ("[\u0531-\u0556][\u0561-\u0586]{2}[\u0561-\u0586]*", [(0x0531, 0x0556, 1, 1), (0x0561, 0x0586, 2, 4)]),                             # Armenian
    ("[\u0905-\u0914][\u0915-\u0939]+[\u093e-\u094c]", [(0x0905, 0x0914, 1, 1), (0x0915, 0x0939, 1, 3), (0x093E, 0x094C, 1, 1)]),  # Devanagari
    ("[\u0e01-\u0e2e][\u0e30-\u0e3a][\u0e01-\u0e2e]", [(0x0E01, 0x0E2E, 1, 1), (0x0E30, 0x0E3A, 1, 1), (0x0E01, 0x0E2E, 1, 1)]),   # Thai
    ("[\u10d0-\u10fa]{4}", [(0x10D0, 0x10FA, 4, 4)]),                                                                       # Georgian
    ("[\u3041-\u3096]+[\u30a1-\u30fa]{2}", [(0x3041, 0x3096, 1, 4), (0x30A1, 0x30FA, 2, 2)]),                              # Hiragana, then Katakana
    ("[\u4e00-\u4e3f][\u4e80-\u4ebf][\u4f00-\u4f3f]", [(0x4E00, 0x4E3F, 1, 1), (0x4E80, 0x4EBF, 1, 1), (0x4F00, 0x4F3F, 1, 1)]),   # three CJK blocks in order
    ("[\u5000-\u503f]{2}[\u5100-\u513f]+[\u5200-\u523f]", [(0x5000, 0x503F, 2, 2), (0x5100, 0x513F, 1, 3), (0x5200, 0x523F, 1, 1)]),
    ("[\uac00-\uac7f]{2}[\uad00-\uad7f]", [(0xAC00, 0xAC7F, 2, 2), (0xAD00, 0xAD7F, 1, 1)]),                              # Hangul
    ("[0-9]+[A-Z][a-z]{2}\u00e9", [(0x30, 0x39, 1, 3), (0x41, 0x5A, 1, 1), (0x61, 0x7A, 2, 2), (0xE9, 0xE9, 1, 1)]),
]


def scriptseq_regex():
    """C5w regex: the union of SEQ_ALTS."""
    return "|".join(a for a, _ in SEQ_ALTS)


def scriptseq_instances(n=256, seed=SEED):
    """n matching strings (lists of code units), round-robin over the alternatives, lengths / chars hashed."""
    out = []
    for i in range(n):
        _, els = SEQ_ALTS[i % len(SEQ_ALTS)]
        w = []
        for j, (lo, hi, mn, mx) in enumerate(els):
            k = mn + _h32i((seed ^ 0x5E9) + i * 131 + j * 7) % (mx - mn + 1)
            w += [lo + _h32i(seed + i * 1009 + j * 101 + t * 1000003) % (hi - lo + 1) for t in range(k)]
        out.append(w)
    return out


def scriptseq_batch(xp, row0, n_rows, n_cols=256, seed=SEED, device=None):
    """C5w: UTF-16 code units: 60 % ASCII letters / digits / space, 10 % BMP symbols outside every range, 30 % chars drawn from the
    alternatives' own ranges (so walks keep entering alternatives and falling out of them); in 30 % of the rows one matching
    instance is planted."""
    r, c = _grid(xp, row0, n_rows, n_cols, device)
    h = _hash32(seed + 55 + r * 1315423911 + c * 2654435761)
    sel = h % 100
    ascii_ch = _lut(xp, ALPHA_C2, device)[(h >> 8) % len(ALPHA_C2)]
    rng = [(lo, hi) for _, els in SEQ_ALTS for lo, hi, _, _ in els if lo > 0x7F and hi > lo]
    starts = _lut(xp, [a for a, _ in rng], device)
    sizes = _lut(xp, [b - a + 1 for a, b in rng], device)
    ri = (h >> 10) % len(rng)
    in_range = starts[ri] + (h >> 17) % sizes[ri]
    other = 0x2000 + (h >> 9) % 0x0C00
    base = xp.where(sel < 60, ascii_ch, xp.where(sel < 70, other, in_range))
    inst = scriptseq_instances()
    maxlen = max(len(w) for w in inst)
    wtab = np.zeros((len(inst), maxlen), dtype=np.int64)
    wlen = np.zeros(len(inst), dtype=np.int64)
    for i, w in enumerate(inst):
        wtab[i, :len(w)] = w
        wlen[i] = len(w)
    wtab, wlen = _lut(xp, wtab, device), _lut(xp, wlen, device)
    hr = _hash32((seed ^ 0x5C5C5C5C) + r * 40503)
    plant = (hr % 10) < 3
    k = (hr >> 4) % len(inst)
    ln = wlen[k]
    pos = (hr >> 12) % (n_cols - ln + 1)
    rel = c - pos
    inside = plant & (rel >= 0) & (rel < ln)
    relc = xp.clip(rel, 0, maxlen - 1) if xp is np else rel.clamp(0, maxlen - 1)
    planted = wtab[k + 0 * c, relc]
    out = xp.where(inside, planted, base)
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
