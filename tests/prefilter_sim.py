"""Host-side restatement of what needle_amd/csrc/needle_ngram.hip computes (the n-gram candidate filter), on the
REFERENCE-layout tables: test one hashed 4-byte window every S chars; where one passes, run the automaton from the start
state K chars ahead of the window's end for K + S - 1 chars, on after a first accept until it dies; a row's answer is the run
that accepted first.  Plain Python, small inputs -- the checker's view of the algorithm, not the product."""
import numpy as np


def window_passes(info, bitmap, x):
    u = ((x & 0xFFFF) * info["m1"] + (x >> 16) * info["m2"]) & 0xFFFFFFFF  # (needle_ngram.h: one v_dot2_u32_u16)
    w = int(bitmap[(u & info["addr_mask"]) >> 2])  # a window owns two bits of one word
    return (w >> ((u >> info["addr_shift"]) & 31)) & (w >> ((u >> (info["addr_shift"] - 8)) & 31)) & 1


def window2_passes(info, bitmap2, x, c5):
    """The second level (needle_ngram.h ngram_probe2): the byte in front of the window joins the hash."""
    u = ((x & 0xFFFF) * info["m1"] + (x >> 16) * info["m2"] + c5 * info["m3"]) & 0xFFFFFFFF
    w = int(bitmap2[(u & info["addr_mask2"]) >> 2])
    return (w >> ((u >> 24) & 31)) & (w >> ((u >> 16) & 31)) & 1


def _probe(info, bitmap, u, mask_key, sh_hi=None):
    w = int(bitmap[(u & info[mask_key]) >> 2])
    hi = info["addr_shift"] if sh_hi is None else sh_hi
    return (w >> ((u >> hi) & 31)) & (w >> ((u >> (hi - 8)) & 31)) & 1


def candidate(info, t, e, all_windows):
    """Does the window ending at e make a candidate: its 4 bytes in the bitmap and -- find / containedIn, 5 chars or more into the row --
    its 5 bytes in the second one.  info["wide"]: the windows are four 16-bit code units (needle_ngram.h ngram_piece16)."""
    if all_windows:
        return True
    if info.get("wide"):
        x0, x1 = int(t[e - 4] | (t[e - 3] << 16)), int(t[e - 2] | (t[e - 1] << 16))
        u = ((x0 & 0xFFFF) * info["m1"] + (x0 >> 16) * info["m2"] + (x1 & 0xFFFF) * info["m1b"] + (x1 >> 16) * info["m2b"]) & 0xFFFFFFFF
        if not _probe(info, info["bitmap"], u, "addr_mask"):
            return False
        if info.get("on2") and e >= 5 and not info.get("no_level2"):
            return bool(_probe(info, info["bitmap2"], (u + int(t[e - 5]) * info["m3"]) & 0xFFFFFFFF, "addr_mask2", 24))
        return True
    x = int(t[e - 4] | (t[e - 3] << 8) | (t[e - 2] << 16) | (t[e - 1] << 24))
    if not window_passes(info, info["bitmap"], x):
        return False
    if info.get("on2") and e >= 5 and not info.get("no_level2"):
        if window2_passes(info, info["bitmap2"], x, int(t[e - 5])):
            return True
        # two-sided (NgramParams::on2 == 2): or the 5 chars ending one char behind the window -- chars [e - 4, e + 1)
        if info["on2"] == 2 and e < len(t):
            xf = int(t[e - 3] | (t[e - 2] << 8) | (t[e - 1] << 16) | (t[e] << 24))
            return bool(window2_passes(info, info["bitmap2"], xf, int(t[e - 4])))
        return False
    return True


class Automaton:
    """step(state, char) -> state (-1 dead), in reference numbering (0 = start)."""

    def __init__(self, table, accepting, class_map, max_char, over=None, dead_to_start=False, absorbing=False, n_dead=0, pend=None):
        self.table, self.acc, self.cm, self.max_char = table, accepting, class_map, max_char
        self.over, self.dead_to_start, self.absorbing, self.n_dead, self.pend = over, dead_to_start, absorbing, n_dead, pend

    def step(self, s, c):
        if self.absorbing and self.acc[s]:
            return s
        if c > self.max_char:
            t = -1 if self.over is None else int(self.over[s])
        else:
            t = int(self.table[s, self.cm[c]])
        if t < 0 and self.dead_to_start:
            t = 0
        return t

    def dead(self, s):
        return s < 0 or 1 <= s <= self.n_dead


def _acc(d):
    a = np.zeros(d["n_states"], dtype=bool)
    a[d["accepting"]] = True  # (Pattern.tables() lists the accepting state ids)
    return a


def from_pattern(p, op):
    """The automaton run_dev puts behind the filter: containedIn's, the lengths automaton, or (one-length patterns) indexForwards'."""
    t = p.tables()
    cm = t["class_map"]
    if op == "contained_in":
        d = t["dfas"]["contained_in"]
        tab = np.asarray(d["table"]).reshape(-1, t["stride"])
        return Automaton(tab, _acc(d), cm, d["max_char"], dead_to_start=True, absorbing=True), None
    if t["fixed_len"] >= 0:
        d = t["dfas"]["forwards"]
        tab = np.asarray(d["table"]).reshape(-1, t["stride"])
        return Automaton(tab, _acc(d), cm, d["max_char"]), t["fixed_len"]
    ml = p.match_length_automaton()
    assert ml is not None
    return Automaton(ml["table"][:, :-1], ml["accepting"], cm, ml["max_char"], over=ml["table"][:, -1], n_dead=ml["n_dead"], pend=ml["pend"]), None


def run_window(au, text, qn, K, S, fixed_len, contained):
    """One candidate (window [qn - 4, qn)): -> None | (first, last, start)."""
    n = len(text)
    r = max(qn - K, 0)
    lim = min(qn + S - 1, n)
    st, pos, first, last = 0, r, None, None
    while pos < lim:
        st = au.step(st, int(text[pos]))
        pos += 1
        if st >= 0 and au.acc[st] and pos >= qn:
            if contained:
                return (pos, pos, 0)
            if first is None:
                first, lim = pos, n
            last = pos
        if au.dead(st):
            break  # (after an accept: possibly one before qn, which is another window's to report)
    if first is None:
        return None
    if fixed_len is not None:
        return (first, last, last - fixed_len)
    if not au.dead(st):  # the row ended with the automaton alive: the END transition leads to the D_L of the pending length
        L = int(au.pend[st])
    else:
        L = int(au.pend[st]) if st > 0 else 0
    assert L > 0
    return (first, last, last - L)


def filtered(p, op, text, all_windows=False, info=None, phase=0):
    """(found, start, end) of one row by the filter algorithm; all_windows: every sampled window counts as a candidate.  phase: the sampled
    window ends are = phase (mod S) (the kernel's are multiples of S; the algorithm does not depend on it)."""
    info = info or p.prefilter_info("contained_in" if op == "contained_in" else "forwards", with_bitmap=True)
    assert info["on"], info
    au, fixed = from_pattern(p, op)
    S, K = info["stride"], info["warm"]
    best = None
    t = np.asarray(text).astype(np.int64)
    for e in range(S + phase % S, len(t) + 1, S):
        if e < 4:
            continue
        if not candidate(info, t, e, all_windows):
            continue
        rep = run_window(au, t, e, K, S, fixed, op == "contained_in")
        if rep is not None and (best is None or rep < best):
            best = rep
    if best is None:
        return (False, -1, -1)
    return (True, best[2], best[1]) if op != "contained_in" else (True, -1, -1)


# ---- find-all behind the filter (needle_ngram.hip, ngram_kernel<OP_NG_FIND_ALL>) ------------------------------------------------
def walk_row_sim(au, text, qn, r, lim0, fixed_len):
    """needle_ngram.hip walk_row: the automaton from the start state at char r, a FIRST accept counted at end indexes >= qn and
    < lim0 + 1, then on until it dies -> dict(found, died, crossed, first, last, start).  crossed: an accepting state BEFORE qn."""
    n = len(text)
    lim = min(lim0, n)
    st, pos, first, last, died, crossed = 0, r, None, None, False, False
    while pos < lim:
        st = au.step(st, int(text[pos]))
        pos += 1
        if st >= 0 and au.acc[st]:
            if pos >= qn:
                if first is None:
                    first, lim = pos, n
                last = pos
            else:
                crossed = True
        if au.dead(st):
            died = True
            break
    h = dict(found=first is not None, died=died and first is None, crossed=crossed, first=first, last=last, start=None)
    if h["found"]:
        if fixed_len is not None:
            h["start"] = last - fixed_len
        else:
            L = int(au.pend[st]) if st > 0 else 0
            assert L > 0
            h["start"] = last - L
    return h


def filtered_find_all(p, text, info=None, all_windows=False, row_slots=2, with_crossed=True, phase=0):
    """Every match of one row by the filter kernel's find-all logic: verified candidates are filed with their row (found, or died /
    crossed an earlier accept on the way: "unknown"), and the row resolves them in window order against its moving cursor; rows with
    more than row_slots entries, and re-runs that cross an unfiled match, take the exact match-by-match loop.  with_crossed=False: the
    round-4 logic (kept to show the case it loses)."""
    info = info or p.prefilter_info("forwards", with_bitmap=True)
    assert info["on"], info
    au, fixed = from_pattern(p, "find")
    S, K = info["stride"], info["warm"]
    t = np.asarray(text).astype(np.int64)
    entries = []
    for e in range(S + phase % S, len(t) + 1, S):
        if e < 4:
            continue
        if not candidate(info, t, e, all_windows):  # (the find-all form asks the second level too where its LDS has room: exact either way)
            continue
        h = walk_row_sim(au, t, e, max(e - K, 0), e + S - 1, fixed)
        crossed = h["crossed"] and with_crossed
        if h["found"] or h["died"] or crossed:
            entries.append((e, (not h["found"]) or crossed, h["last"], h["start"]))
    out, cursor = [], 0
    slow = len(entries) > row_slots
    if not slow:
        for e, unknown, last, start in entries:
            if not unknown and max(e - K, 0) >= cursor:
                out.append((start, last))
                cursor = last
            elif e + S - 1 > cursor:
                h = walk_row_sim(au, t, max(e, cursor + 1), cursor, e + S - 1, fixed)
                if h["crossed"] and with_crossed:
                    slow = True
                    break
                if h["found"]:
                    out.append((h["start"], h["last"]))
                    cursor = h["last"]
    while slow:
        h = walk_row_sim(au, t, cursor, cursor, 1 << 30, fixed)
        if not h["found"]:
            break
        out.append((h["start"], h["last"]))
        cursor = h["last"]
    return out
