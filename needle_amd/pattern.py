"""Host-side mirror of the reference's interface for this path, over the C ABI (include/needle_hip.h):

    DFACompiler.compile(regex, className[, flags]) -> Pattern   needle-compiler/.../DFACompiler.java:16-37
    Pattern.matcher(String) -> Matcher, flag constants          needle-types/.../Pattern.java:3-34
    Matcher.matches/containedIn/find/find(int,int)/start/end    needle-types/.../Matcher.java:6-26

plus the batch entry points the reference does not have (one launch over many haystacks).  Everything that
matches goes through the HIP kernels; nothing here walks an automaton on the CPU.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import BatchView, DfaDesc, PatternInfo, TableDesc

# flag constants, same values as the reference (Pattern.java:9-31)
CASE_INSENSITIVE = 0x02
DOTALL = 0x20
UNICODE_CASE = 0x40
UNICODE_CHARACTER_CLASS = 0x100
LEFTMOST_LONGEST = 0x800000
ALL_FLAGS = DOTALL | CASE_INSENSITIVE | UNICODE_CASE | LEFTMOST_LONGEST | UNICODE_CHARACTER_CLASS


class PatternException(RuntimeError):
    """NC/PatternException.java"""


class PatternSyntaxException(PatternException):
    """NC/PatternSyntaxException.java (RegexParser.java:86-98,293-295)"""


class PatternClassCompilationException(PatternException):
    """NC/PatternClassCompilationException.java (DFACompiler.java:34-36,71-83)"""


class DeviceError(RuntimeError):
    """HIP failure / no device (IllegalStateException in the Java shim)."""


def _check(rc):
    if rc == _lib.NEEDLE_OK:
        return
    msg = _lib.last_error()
    if rc == _lib.ERR_INVALID:
        raise ValueError(msg)
    if rc == _lib.ERR_SYNTAX:
        raise PatternSyntaxException(msg)
    if rc in (_lib.ERR_COMPILE, _lib.ERR_UNSUPPORTED):
        raise PatternClassCompilationException(msg)
    raise DeviceError(msg)


def _utf16(s):
    """str -> uint16 code units (surrogate pairs for astral chars, like java.lang.String)."""
    if isinstance(s, np.ndarray):
        return np.ascontiguousarray(s, dtype=np.uint16)
    return np.frombuffer(s.encode("utf-16-le", "surrogatepass"), dtype=np.uint16).copy()


class Matcher:
    """One haystack + cursor (the generated class's fields, DFAClassBuilder.java:688-699)."""

    def __init__(self, pattern, s):
        self._pattern = pattern  # keeps the native pattern alive
        self._units = _utf16(s)
        h = ctypes.c_void_p()
        _check(_lib.lib().needle_matcher_create(pattern._h, self._units.ctypes.data, self._units.size, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and _lib._lib is not None:  # module globals may be gone at interpreter exit
            _lib._lib.needle_matcher_destroy(h)

    def _bool(self, fn, *args):
        r = ctypes.c_int(0)
        _check(fn(self._h, *args, ctypes.byref(r)))
        return bool(r.value)

    def matches(self):
        return self._bool(_lib.lib().needle_matcher_matches)

    def containedIn(self):
        return self._bool(_lib.lib().needle_matcher_contained_in)

    def find(self, start=None, end=None):
        if start is None:
            return self._bool(_lib.lib().needle_matcher_find)
        return self._bool(_lib.lib().needle_matcher_find_range, int(start), int(end))

    def start(self):
        return _lib.lib().needle_matcher_start(self._h)

    def end(self):
        return _lib.lib().needle_matcher_end(self._h)


WHICH = {"matches": 0, "contained_in": 1, "forwards": 2, "backwards": 3}


class Pattern:
    """A compiled pattern: immutable, shareable, tables resident in HBM per device."""

    def __init__(self, handle, regex=None, flags=0):
        self._h = handle
        self.regex = regex
        self.flags = flags

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and _lib._lib is not None:
            _lib._lib.needle_pattern_destroy(h)

    # ---- reference interface
    def matcher(self, s):
        return Matcher(self, s)

    # ---- construction from the tables a generated class carries
    @classmethod
    def from_tables(cls, class_map, stride, dfas, fixed_len=-1):
        """dfas: {"matches"|"contained_in"|"forwards"|"backwards": dict(n_states, max_char, accepting (ids),
        table (flat int16) | table_strings (list[str]))}"""
        cm = np.ascontiguousarray(class_map, dtype=np.uint8)
        assert cm.size == 65536
        keep = [cm]
        desc = TableDesc()
        desc.class_map = cm.ctypes.data
        desc.stride = int(stride)
        desc.fixed_len = -1 if fixed_len is None else int(fixed_len)
        for name in WHICH:
            spec = dfas[name]
            d = DfaDesc()
            d.n_states = int(spec["n_states"])
            d.max_char = 0xFFFF if spec.get("max_char") is None else int(spec["max_char"])
            acc = np.zeros(d.n_states, dtype=np.uint8)
            for s in spec["accepting"]:
                acc[int(s)] = 1
            keep.append(acc)
            d.accepting = acc.ctypes.data
            if spec.get("table") is not None:
                t = np.ascontiguousarray(spec["table"], dtype=np.int16)
                assert t.size == d.n_states * desc.stride
                keep.append(t)
                d.table = t.ctypes.data
                d.table_string = None
            else:
                d.table = None
                d.table_string = ";".join(spec["table_strings"]).encode("ascii")
            setattr(desc, name, d)
        h = ctypes.c_void_p()
        _check(_lib.lib().needle_pattern_from_tables(ctypes.byref(desc), ctypes.byref(h)))
        del keep
        return cls(h)

    # ---- precompiled-pattern blob (the analogue of Precompile.precompile, NC/precompile/Precompile.java:19-53)
    def to_bytes(self):
        need = ctypes.c_size_t(0)
        _check(_lib.lib().needle_pattern_serialize(self._h, None, 0, ctypes.byref(need)))
        buf = (ctypes.c_ubyte * need.value)()
        _check(_lib.lib().needle_pattern_serialize(self._h, buf, need.value, ctypes.byref(need)))
        return bytes(buf)

    @classmethod
    def from_bytes(cls, blob):
        h = ctypes.c_void_p()
        b = (ctypes.c_ubyte * len(blob)).from_buffer_copy(blob)
        _check(_lib.lib().needle_pattern_deserialize(b, len(blob), ctypes.byref(h)))
        return cls(h)

    # ---- introspection
    def info(self):
        i = PatternInfo()
        _check(_lib.lib().needle_pattern_get_info(self._h, ctypes.byref(i)))
        return {"stride": i.stride, "n_states": dict(zip(WHICH, i.n_states)), "max_char": dict(zip(WHICH, i.max_char)),
                "fixed_len": i.fixed_len, "min_len": i.min_len, "max_len": i.max_len,
                "kernel_mode": dict(zip(WHICH, i.kernel_mode))}

    def program_info(self, which="forwards", char_width=1, with_backward=True):
        """How one automaton is lowered for the device (host-side diagnostics: needs no GPU)."""
        i = _lib.ProgramInfo()
        _check(_lib.lib().needle_pattern_program_info(self._h, list(WHICH).index(which), char_width, int(with_backward), ctypes.byref(i)))
        return {k: getattr(i, k) for k, _ in i._fields_}

    def prefilter_info(self, which="forwards", with_bitmap=False, wide=False):
        """The n-gram candidate filter behind which containedIn() / find() run on batches of 8-bit rows (SURVEY.md s8 f-4;
        host-side diagnostics: needs no GPU): {"on", "mode", "stride", "warm", "min_len", "n_windows", "bitmap_bytes", hash
        parameters, "why" (what ruled it out)} and, with_bitmap, "bitmap" (uint32 words).  wide: the filter of UTF-16 rows of a
        pattern on several pages of the BMP (windows of four code units: "m1b", "m2b" -- needle_pattern_prefilter_info2)."""
        i2 = _lib.PrefilterInfo2()
        L = _lib.lib()
        wi = list(WHICH).index(which)
        _check(L.needle_pattern_prefilter_info2(self._h, wi, int(bool(wide)), ctypes.byref(i2), None, 0))
        i = i2.base
        out = {k: getattr(i, k) for k, _ in i._fields_}
        out["why"] = out["why"].decode()
        out["wide"], out["m1b"], out["m2b"] = int(i2.wide), int(i2.m1b), int(i2.m2b)
        if with_bitmap and i.on:
            bm = np.zeros((i.bitmap_bytes + (i.bitmap2_bytes if i.on2 else 0)) // 4, dtype=np.uint32)
            _check(L.needle_pattern_prefilter_info2(self._h, wi, int(bool(wide)), ctypes.byref(i2), bm.ctypes.data, bm.size))
            out["bitmap"] = bm[:i.bitmap_bytes // 4]
            if i.on2:
                out["bitmap2"] = bm[i.bitmap_bytes // 4:]  # the second level's (5-byte windows)
        return out

    PREFILTER_AUTO, PREFILTER_ON, PREFILTER_OFF = 0, 1, 2

    def set_prefilter(self, mode):
        """needle_pattern_set_prefilter: AUTO (the flood watch decides), ON (the filter kernel whenever the shape allows, never suspended),
        OFF (the ordinary scan kernels).  Answers are the same in all three."""
        _check(_lib.lib().needle_pattern_set_prefilter(self._h, int(mode)))

    def prefilter_state(self, which="forwards"):
        """needle_pattern_prefilter_state: what the flood watch of this pattern's filter program knows on the current device."""
        st = _lib.PrefilterState()
        _check(_lib.lib().needle_pattern_prefilter_state(self._h, list(WHICH).index(which), ctypes.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def utf16_route(self):
        """needle_pattern_utf16_route: (page, sub) -- UTF-16 rows of this pattern can run behind the byte program of BMP page `page` (0: ASCII /
        Latin-1, 4: Cyrillic ...), every char outside the page narrowed to byte `sub`; None when the pattern spans several pages.  No GPU needed."""
        page, sub = ctypes.c_int32(0), ctypes.c_int32(0)
        _check(_lib.lib().needle_pattern_utf16_route(self._h, ctypes.byref(page), ctypes.byref(sub)))
        return None if page.value < 0 else (page.value, sub.value)

    def match_length_automaton(self):
        """The refined forward automaton behind find-all's "lengths" form (needle_pattern_match_lengths), or None when the
        pattern does not allow it -> {"n_states", "n_dead", "max_char", "table" int16[n, stride + 1] (last column: chars beyond max_char), "accepting"
        bool[n], "pend" uint8[n]}."""
        L = _lib.lib()
        avail, n, nd, mc = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        _check(L.needle_pattern_match_lengths(self._h, ctypes.byref(avail), ctypes.byref(n), ctypes.byref(nd), ctypes.byref(mc), None, None, None))
        if not avail.value:
            return None
        stride = self.info()["stride"]
        table = np.zeros(n.value * (stride + 1), dtype=np.int16)
        acc = np.zeros(n.value, dtype=np.uint8)
        pend = np.zeros(n.value, dtype=np.uint8)
        _check(L.needle_pattern_match_lengths(self._h, ctypes.byref(avail), ctypes.byref(n), ctypes.byref(nd), ctypes.byref(mc), table.ctypes.data,
                                              acc.ctypes.data, pend.ctypes.data))
        return {"n_states": n.value, "n_dead": nd.value, "max_char": mc.value, "table": table.reshape(n.value, stride + 1),
                "accepting": acc.astype(bool), "pend": pend}

    def find_all_transducer(self, char_width=1):
        """The device program of the find-all transducer (needle_pattern_find_all_transducer: lock-step find-all), or None when the
        pattern has none -> {"n_states", "n_cols", "pad_col", "start", "window", "win_lo_e", "win_hi_e", "off_table", "codes_off",
        "lds_bytes", "n_pages", "kind" (1: the transducer on the lengths automaton -- a code names (length, k); 2: the RUN transducer of
        patterns without bounded match lengths -- code bit 0: a match ends in front of this char, bit 1: this char may begin a run),
        "blob": uint8[lds_bytes]}."""
        L = _lib.lib()
        avail, need = ctypes.c_int32(0), ctypes.c_size_t(0)
        info = (ctypes.c_int32 * 12)()
        _check(L.needle_pattern_find_all_transducer(self._h, int(char_width), ctypes.byref(avail), info, None, 0, ctypes.byref(need)))
        if not avail.value:
            return None
        blob = np.zeros(need.value, dtype=np.uint8)
        _check(L.needle_pattern_find_all_transducer(self._h, int(char_width), ctypes.byref(avail), info, blob.ctypes.data, blob.size, ctypes.byref(need)))
        keys = ["n_states", "n_cols", "pad_col", "start", "window", "win_lo_e", "win_hi_e", "off_table", "codes_off", "lds_bytes", "n_pages", "kind"]
        out = {k: int(info[i]) for i, k in enumerate(keys)}
        out["blob"] = blob
        return out

    def tables(self):
        """The pattern's tables in the reference layout (class map, stride, 4 x (table, accepting, max_char))."""
        inf = self.info()
        cm = np.zeros(65536, dtype=np.uint8)
        _check(_lib.lib().needle_pattern_get_class_map(self._h, cm.ctypes.data))
        out = {"class_map": cm, "stride": inf["stride"], "fixed_len": inf["fixed_len"], "min_len": inf["min_len"],
               "max_len": inf["max_len"], "dfas": {}}
        for name, w in WHICH.items():
            n = inf["n_states"][name]
            t = np.zeros(n * inf["stride"], dtype=np.int16)
            a = np.zeros(n, dtype=np.uint8)
            _check(_lib.lib().needle_pattern_get_table(self._h, w, t.ctypes.data, a.ctypes.data))
            out["dfas"][name] = {"n_states": n, "table": t, "accepting": np.nonzero(a)[0].tolist(),
                                 "max_char": inf["max_char"][name]}
        return out

    # ---- batches
    def _run(self, op, rows, lengths, stream, out=None):
        L = _lib.lib()
        v = BatchView()
        if isinstance(rows, np.ndarray):  # host buffers: upload + run + download inside the library
            rows = np.ascontiguousarray(rows)
            if rows.dtype == np.int16:
                rows = rows.view(np.uint16)
            assert rows.ndim == 2 and rows.dtype in (np.uint8, np.uint16), "rows: 2-D uint8/uint16"
            n, stride = rows.shape
            v.rows, v.char_width, v.n_rows, v.row_stride, v.row_len = rows.ctypes.data, rows.dtype.itemsize, n, stride, stride
            if lengths is not None:
                lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
                assert lengths.shape == (n,)
                v.lengths = lengths.ctypes.data
            words = np.zeros((n + 63) // 64, dtype=np.uint64)
            if op == "find":
                st = np.full(n, -1, dtype=np.int32)
                en = np.full(n, -1, dtype=np.int32)
                _check(L.needle_find_host(self._h, ctypes.byref(v), words.ctypes.data, st.ctypes.data, en.ctypes.data))
                return words, st, en
            fn = L.needle_matches_host if op == "matches" else L.needle_contained_in_host
            _check(fn(self._h, ctypes.byref(v), words.ctypes.data))
            return words
        import torch  # device tensors: plumbing only (memory + stream)
        assert isinstance(rows, torch.Tensor) and rows.is_cuda and rows.dim() == 2 and rows.is_contiguous()
        assert rows.dtype in (torch.uint8, torch.int16, torch.uint16), "rows: uint8 or (u)int16 code units"
        n, stride = rows.shape
        cw = rows.element_size()
        v.rows, v.char_width, v.n_rows, v.row_stride, v.row_len = rows.data_ptr(), cw, n, stride, stride
        if lengths is not None:
            assert lengths.is_cuda and lengths.dtype == torch.int32 and lengths.shape == (n,) and lengths.is_contiguous()
            v.lengths = lengths.data_ptr()
        with torch.cuda.device(rows.device):
            s = torch.cuda.current_stream(rows.device).cuda_stream if stream is None else stream
            if out is not None:  # caller-owned result buffers (at least as large as the results): no allocation per call
                words = out[0] if isinstance(out, (tuple, list)) else out
                assert words.is_cuda and words.dtype == torch.int64 and words.numel() >= (n + 63) // 64
            else:
                words = torch.empty((n + 63) // 64, dtype=torch.int64, device=rows.device)
            if op == "find":
                if out is not None:
                    st, en = out[1], out[2]
                    assert st.dtype == torch.int32 and en.dtype == torch.int32 and st.numel() >= n and en.numel() >= n
                else:
                    st = torch.empty(n, dtype=torch.int32, device=rows.device)
                    en = torch.empty(n, dtype=torch.int32, device=rows.device)
                _check(L.needle_find_dev(self._h, ctypes.byref(v), words.data_ptr(), st.data_ptr(), en.data_ptr(), s))
                return words, st, en
            fn = L.needle_matches_dev if op == "matches" else L.needle_contained_in_dev
            _check(fn(self._h, ctypes.byref(v), words.data_ptr(), s))
            return words

    def matches_batch(self, rows, lengths=None, stream=None, out=None):
        """bitmap words (bit r&63 of word r>>6 = matches() of row r).  out: optional caller-owned int64 device tensor
        for the bitmap words (device batches only)."""
        return self._run("matches", rows, lengths, stream, out)

    def contained_in_batch(self, rows, lengths=None, stream=None, out=None):
        return self._run("contained_in", rows, lengths, stream, out)

    def find_next_batch(self, rows, cursor, lengths=None, stream=None):
        """One Matcher.find() step per row from the per-row cursor (int32 device tensor; < 0 = exhausted).
        -> (bitmap words, start, end); `end` is the rows' next cursor (-1 where nothing was found)."""
        import torch
        L = _lib.lib()
        assert isinstance(rows, torch.Tensor) and rows.is_cuda and rows.dim() == 2 and rows.is_contiguous()
        assert cursor.is_cuda and cursor.dtype == torch.int32 and cursor.shape == (rows.shape[0],) and cursor.is_contiguous()
        v = BatchView()
        n, stride = rows.shape
        v.rows, v.char_width, v.n_rows, v.row_stride, v.row_len = rows.data_ptr(), rows.element_size(), n, stride, stride
        if lengths is not None:
            assert lengths.is_cuda and lengths.dtype == torch.int32 and lengths.shape == (n,)
            v.lengths = lengths.data_ptr()
        with torch.cuda.device(rows.device):
            s = torch.cuda.current_stream(rows.device).cuda_stream if stream is None else stream
            words = torch.empty((n + 63) // 64, dtype=torch.int64, device=rows.device)
            st = torch.empty(n, dtype=torch.int32, device=rows.device)
            en = torch.empty(n, dtype=torch.int32, device=rows.device)
            _check(L.needle_find_next_dev(self._h, ctypes.byref(v), cursor.data_ptr(), words.data_ptr(), st.data_ptr(),
                                          en.data_ptr(), s))
        return words, st, en

    def find_all_dense(self, rows, max_per_row, lengths=None, stream=None, out=None):
        """needle_find_all_dev: every non-overlapping match of every row in dense per-row slots.
        -> (counts int32[n], start int32[n, max_per_row], end int32[n, max_per_row], more: bool)
        out: optional caller-owned (counts, start, end) device tensors (slots beyond a row's count are left as they are)."""
        L = _lib.lib()
        if isinstance(rows, np.ndarray):  # host buffers: needle_find_all_host
            rows = np.ascontiguousarray(rows)
            if rows.dtype == np.int16:
                rows = rows.view(np.uint16)
            assert rows.ndim == 2 and rows.dtype in (np.uint8, np.uint16)
            n, stride = rows.shape
            v = BatchView()
            v.rows, v.char_width, v.n_rows, v.row_stride, v.row_len = rows.ctypes.data, rows.dtype.itemsize, n, stride, stride
            if lengths is not None:
                lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
                v.lengths = lengths.ctypes.data
            counts = np.zeros(n, dtype=np.uint32)
            st = np.full((n, max_per_row), -1, dtype=np.int32)
            en = np.full((n, max_per_row), -1, dtype=np.int32)
            more = ctypes.c_int(0)
            _check(L.needle_find_all_host(self._h, ctypes.byref(v), int(max_per_row), counts.ctypes.data, st.ctypes.data,
                                          en.ctypes.data, ctypes.byref(more)))
            return counts, st, en, bool(more.value)
        import torch
        assert isinstance(rows, torch.Tensor) and rows.is_cuda and rows.dim() == 2 and rows.is_contiguous()
        v = BatchView()
        n, stride = rows.shape
        v.rows, v.char_width, v.n_rows, v.row_stride, v.row_len = rows.data_ptr(), rows.element_size(), n, stride, stride
        if lengths is not None:
            assert lengths.is_cuda and lengths.dtype == torch.int32 and lengths.shape == (n,)
            v.lengths = lengths.data_ptr()
        with torch.cuda.device(rows.device):
            s = torch.cuda.current_stream(rows.device).cuda_stream if stream is None else stream
            if out is not None:
                counts, st, en = out
                assert counts.shape == (n,) and st.shape == (n, max_per_row) and en.shape == (n, max_per_row)
                assert all(t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() for t in out)
            else:
                counts = torch.zeros(n, dtype=torch.int32, device=rows.device)
                st = torch.full((n, max_per_row), -1, dtype=torch.int32, device=rows.device)
                en = torch.full((n, max_per_row), -1, dtype=torch.int32, device=rows.device)
            more = ctypes.c_int(0)
            _check(L.needle_find_all_dev(self._h, ctypes.byref(v), int(max_per_row), counts.data_ptr(), st.data_ptr(),
                                         en.data_ptr(), ctypes.byref(more), s))
        return counts, st, en, bool(more.value)

    def find_all_dense_packed16(self, rows, max_per_row, lengths=None, stream=None, out=None, want_more=True):
        """needle_find_all_packed16_dev: as find_all_dense on device rows of at most 65 535 chars, each match one dword
        (start | end << 16).  -> (counts int32[n], start_end16 int32[n, max_per_row] (bit pattern of the uint32), more: bool or
        None when want_more is False -- the call is then asynchronous on the stream)."""
        L = _lib.lib()
        if isinstance(rows, np.ndarray):  # host buffers: needle_find_all_packed16_host
            rows = np.ascontiguousarray(rows)
            if rows.dtype == np.int16:
                rows = rows.view(np.uint16)
            assert rows.ndim == 2 and rows.dtype in (np.uint8, np.uint16)
            n, stride = rows.shape
            v = BatchView()
            v.rows, v.char_width, v.n_rows, v.row_stride, v.row_len = rows.ctypes.data, rows.dtype.itemsize, n, stride, stride
            if lengths is not None:
                lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
                v.lengths = lengths.ctypes.data
            counts = np.zeros(n, dtype=np.uint32)
            se = np.full((n, max_per_row), 0xFFFFFFFF, dtype=np.uint32)
            more = ctypes.c_int(0)
            _check(L.needle_find_all_packed16_host(self._h, ctypes.byref(v), int(max_per_row), counts.ctypes.data, se.ctypes.data, ctypes.byref(more)))
            return counts, se, bool(more.value)
        import torch
        v, n = self._dev_view(rows, lengths), rows.shape[0]
        with torch.cuda.device(rows.device):
            s = torch.cuda.current_stream(rows.device).cuda_stream if stream is None else stream
            if out is not None:
                counts, se = out
                assert counts.shape == (n,) and se.shape == (n, max_per_row)
                assert all(t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() for t in out)
            else:
                counts = torch.zeros(n, dtype=torch.int32, device=rows.device)
                se = torch.full((n, max_per_row), -1, dtype=torch.int32, device=rows.device)
            more = ctypes.c_int(0)
            _check(L.needle_find_all_packed16_dev(self._h, ctypes.byref(v), int(max_per_row), counts.data_ptr(), se.data_ptr(),
                                                  ctypes.byref(more) if want_more else None, s))
        return counts, se, (bool(more.value) if want_more else None)

    def find_all_blocked16(self, rows, max_per_row, lengths=None, stream=None, out=None, want_more=True):
        """needle_find_all_blocked16_dev: find_all_dense_packed16 with GROUP-BLOCKED slots -- match k of row r at
        blocks[r >> 6, k, r & 63].  -> (counts int32[n], blocks int32[ceil(n / 64), max_per_row, 64], more)
        (unblock16(blocks, n) gives the [n, max_per_row] view of find_all_dense_packed16)."""
        import torch
        L = _lib.lib()
        v, n = self._dev_view(rows, lengths), rows.shape[0]
        ng = (n + 63) // 64
        with torch.cuda.device(rows.device):
            s = torch.cuda.current_stream(rows.device).cuda_stream if stream is None else stream
            if out is not None:
                counts, blocks = out
                assert counts.shape == (n,) and blocks.shape == (ng, max_per_row, 64)
                assert all(t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() for t in out)
            else:
                counts = torch.zeros(n, dtype=torch.int32, device=rows.device)
                blocks = torch.full((ng, max_per_row, 64), -1, dtype=torch.int32, device=rows.device)
            more = ctypes.c_int(0)
            _check(L.needle_find_all_blocked16_dev(self._h, ctypes.byref(v), int(max_per_row), counts.data_ptr(), blocks.data_ptr(),
                                                   ctypes.byref(more) if want_more else None, s))
        return counts, blocks, (bool(more.value) if want_more else None)

    def find_all_compact16(self, rows, max_per_row=32, lengths=None, stream=None, cap=None, want_more=True):
        """needle_find_all_compact16_dev: every match of every row in compact (CSR) form in ONE call that walks the text once
        -> (offsets int64[n + 1], start_end16 int32[total] (start | end << 16), more: bool or None).  cap: dwords of room for the matches
        (default: 8 per row, grown to the exact total when that was too little); more = some row had more than max_per_row matches (its
        list is cut: raise max_per_row, or use find_all_csr)."""
        import torch
        L = _lib.lib()
        v, n = self._dev_view(rows, lengths), rows.shape[0]
        with torch.cuda.device(rows.device):
            s = torch.cuda.current_stream(rows.device).cuda_stream if stream is None else stream
            offsets = torch.empty(n + 1, dtype=torch.int64, device=rows.device)
            total = torch.zeros(1, dtype=torch.int64, device=rows.device)
            cap = int(cap) if cap is not None else max(1024, 8 * n)
            while True:
                se = torch.empty(cap, dtype=torch.int32, device=rows.device)
                more = ctypes.c_int(0)
                _check(L.needle_find_all_compact16_dev(self._h, ctypes.byref(v), int(max_per_row), offsets.data_ptr(), se.data_ptr(), cap,
                                                       total.data_ptr(), ctypes.byref(more) if want_more else None, s))
                t = int(total.item())
                if t <= cap:
                    return offsets, se[:t], (bool(more.value) if want_more else None)
                cap = t

    @staticmethod
    def unblock16(blocks, n_rows):
        """[groups, slots, 64] group-blocked slots -> [n_rows, slots] row-major (a copy)."""
        g, k, _ = blocks.shape
        return blocks.permute(0, 2, 1).reshape(g * 64, k)[:n_rows].contiguous()

    def _dev_view(self, rows, lengths):
        import torch
        assert isinstance(rows, torch.Tensor) and rows.is_cuda and rows.dim() == 2 and rows.is_contiguous()
        v = BatchView()
        n, stride = rows.shape
        v.rows, v.char_width, v.n_rows, v.row_stride, v.row_len = rows.data_ptr(), rows.element_size(), n, stride, stride
        if lengths is not None:
            assert lengths.is_cuda and lengths.dtype == torch.int32 and lengths.shape == (n,)
            v.lengths = lengths.data_ptr()
        return v

    def count_matches_batch(self, rows, lengths=None, stream=None):
        """needle_count_matches_dev: int32[n_rows], the number of non-overlapping matches of every row."""
        import torch
        L = _lib.lib()
        v = self._dev_view(rows, lengths)
        with torch.cuda.device(rows.device):
            s = torch.cuda.current_stream(rows.device).cuda_stream if stream is None else stream
            counts = torch.empty(rows.shape[0], dtype=torch.int32, device=rows.device)
            _check(L.needle_count_matches_dev(self._h, ctypes.byref(v), counts.data_ptr(), s))
        return counts

    def find_all_csr(self, rows, lengths=None, stream=None):
        """Every non-overlapping match of every row in compact form, two passes over the batch: count
        (needle_count_matches_dev), exclusive prefix sum, fill (needle_find_all_csr_dev).
        -> (offsets int64[n_rows + 1], start int32[m], end int32[m])"""
        L = _lib.lib()
        if isinstance(rows, np.ndarray):  # host buffers: needle_find_all_csr_host (first with a guessed capacity)
            rows = np.ascontiguousarray(rows)
            if rows.dtype == np.int16:
                rows = rows.view(np.uint16)
            assert rows.ndim == 2 and rows.dtype in (np.uint8, np.uint16)
            n, stride = rows.shape
            v = BatchView()
            v.rows, v.char_width, v.n_rows, v.row_stride, v.row_len = rows.ctypes.data, rows.dtype.itemsize, n, stride, stride
            if lengths is not None:
                lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
                v.lengths = lengths.ctypes.data
            offsets = np.zeros(n + 1, dtype=np.uint64)
            capacity = max(1024, 2 * n)
            while True:
                st = np.empty(capacity, dtype=np.int32)
                en = np.empty(capacity, dtype=np.int32)
                total = ctypes.c_uint64(0)
                _check(L.needle_find_all_csr_host(self._h, ctypes.byref(v), offsets.ctypes.data, st.ctypes.data, en.ctypes.data, capacity,
                                                  ctypes.byref(total)))
                if total.value <= capacity:
                    return offsets.astype(np.int64), st[:total.value], en[:total.value]
                capacity = int(total.value)
        import torch
        v = self._dev_view(rows, lengths)
        n = rows.shape[0]
        # the count pass, the prefix sum (torch) and the fill pass all run on ONE stream: a caller-supplied raw stream is made
        # torch's current stream for the torch ops in between (else the cumsum could read counts the count kernel has not
        # written yet, and the fill pass offsets the cumsum has not)
        dev = rows.device
        one = torch.cuda.current_stream(dev) if stream is None else torch.cuda.ExternalStream(stream, device=dev)
        with torch.cuda.device(dev), torch.cuda.stream(one):
            s = one.cuda_stream
            counts = self.count_matches_batch(rows, lengths, s)
            offsets = torch.zeros(n + 1, dtype=torch.int64, device=rows.device)
            torch.cumsum(counts, 0, out=offsets[1:])
            total = int(offsets[-1].item())
            st = torch.empty(total, dtype=torch.int32, device=rows.device)
            en = torch.empty(total, dtype=torch.int32, device=rows.device)
            if total == 0:
                return offsets, st, en
            more = ctypes.c_int(0)
            _check(L.needle_find_all_csr_dev(self._h, ctypes.byref(v), offsets.data_ptr(), st.data_ptr(), en.data_ptr(), ctypes.byref(more), s))
            assert not more.value, "count pass and fill pass disagree"
        return offsets, st, en

    def find_all_batch(self, rows, lengths=None, max_rounds=None):
        """Every non-overlapping match of every row, as the reference's repeated find() would report them
        (DFACompilerTest.java:66-78,671-699) -> (offsets int64[n_rows + 1], start int32[m], end int32[m]) in CSR form:
        find_all_csr (count pass, prefix sum, fill pass); with max_rounds, or for rows of 64 MiB and more, one
        find_next launch per round over the rows that still have a cursor.
        A row ends where the reference's cursor stops advancing: an EMPTY match (or any match that does not end beyond
        the cursor it was searched from) is reported once and ends its row; a nullable pattern's wrapped pseudo-match
        at cursor == length (start = length, end = 0), on which the reference would cycle for ever, is dropped."""
        import torch
        n = rows.shape[0]
        dev = rows.device
        if max_rounds is None and rows.shape[1] * rows.element_size() < (1 << 26):  # count, prefix sum, fill: two passes
            return self.find_all_csr(rows, lengths)
        cursor = torch.zeros(n, dtype=torch.int32, device=dev)
        ids = torch.arange(n, dtype=torch.int64, device=dev)
        counts = torch.zeros(n, dtype=torch.int64, device=dev)
        per_round = []  # round k holds the k-th match of every row that has one: no sort needed to build the CSR
        round_cap = rows.shape[1] + 1  # a row of L chars has at most L non-empty matches (+ one empty one)
        while len(per_round) < round_cap:
            _, st, en = self.find_next_batch(rows, cursor, lengths)
            # en < st: the reference's wrapped pseudo-match of a nullable pattern searched from cursor == length
            # (end = the literal 0 of DFAClassBuilder.java:356); dropped, it ends the row (see needle_find_all_dev)
            hit = (en >= 0) & (en >= st) & (cursor >= 0)
            if not bool(hit.any()):
                break
            per_round.append((ids[hit], st[hit], en[hit]))
            counts += hit
            stop = (en == st) | (en <= cursor)  # the row goes on only while the cursor advances
            cursor = torch.where(hit & ~stop, en, torch.full_like(en, -1))
            if max_rounds is not None and len(per_round) >= max_rounds:
                break
        offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        offsets[1:] = torch.cumsum(counts, 0)
        total = int(offsets[-1].item())
        out_s = torch.empty(total, dtype=torch.int32, device=dev)
        out_e = torch.empty(total, dtype=torch.int32, device=dev)
        for k, (r, s_, e_) in enumerate(per_round):
            pos = offsets[r] + k
            out_s[pos] = s_
            out_e[pos] = e_
        return offsets, out_s, out_e

    def find_batch(self, rows, lengths=None, stream=None, out=None):
        """(bitmap words, start int32[n], end int32[n]); unmatched rows have start = end = -1.  out: optional
        caller-owned (bitmap int64, start int32, end int32) device tensors."""
        return self._run("find", rows, lengths, stream, out)

    def find_packed16_batch(self, rows, lengths=None, stream=None, out=None):
        """needle_find_packed16_dev: find() on device rows of at most 65 534 chars -> (bitmap words, int32[n] tensor whose
        elements are the dwords start | end << 16, -1 (0xFFFFFFFF) = no match), stored by the scan kernel itself: 4 result bytes
        per row instead of 8.  out: optional caller-owned (bitmap int64, packed int32) device tensors."""
        import torch
        v = self._dev_view(rows, lengths)
        n = rows.shape[0]
        with torch.cuda.device(rows.device):
            s = torch.cuda.current_stream(rows.device).cuda_stream if stream is None else stream
            if out is not None:
                words, se = out
                assert words.dtype == torch.int64 and words.numel() >= (n + 63) // 64 and se.dtype == torch.int32 and se.numel() >= n
            else:
                words = torch.empty((n + 63) // 64, dtype=torch.int64, device=rows.device)
                se = torch.empty(n, dtype=torch.int32, device=rows.device)
            _check(_lib.lib().needle_find_packed16_dev(self._h, ctypes.byref(v), words.data_ptr(), se.data_ptr(), s))
        return words, se

    def find_packed8_batch(self, rows, lengths=None, stream=None, out=None):
        """needle_find_packed8_dev: find() on device rows of at most 256 chars -> (bitmap words, int16[n] tensor whose elements are the
        uint16 start | (end - start) << 8; 0xFFFF = no match, 0xFFFE = the match (0, 256)): 2 result bytes per row.
        out: optional caller-owned (bitmap int64, packed int16) device tensors.  unpack8() decodes."""
        import torch
        v = self._dev_view(rows, lengths)
        n = rows.shape[0]
        with torch.cuda.device(rows.device):
            s = torch.cuda.current_stream(rows.device).cuda_stream if stream is None else stream
            if out is not None:
                words, sl = out
                assert words.dtype == torch.int64 and words.numel() >= (n + 63) // 64 and sl.dtype == torch.int16 and sl.numel() >= n
            else:
                words = torch.empty((n + 63) // 64, dtype=torch.int64, device=rows.device)
                sl = torch.empty(n, dtype=torch.int16, device=rows.device)
            _check(_lib.lib().needle_find_packed8_dev(self._h, ctypes.byref(v), words.data_ptr(), sl.data_ptr(), s))
        return words, sl

    @staticmethod
    def unpack8(sl):
        """uint16 entries of needle_find_packed8_* (numpy array, any integer dtype holding the bit pattern) -> (start, end) int32
        arrays, -1 / -1 where there is no match."""
        x = np.asarray(sl).astype(np.int64) & 0xFFFF
        start = np.where(x == 0xFFFF, -1, np.where(x == 0xFFFE, 0, x & 0xFF))
        end = np.where(x == 0xFFFF, -1, np.where(x == 0xFFFE, 256, (x & 0xFF) + (x >> 8)))
        return start.astype(np.int32), end.astype(np.int32)

    def find_packed8_host(self, rows, lengths=None):
        """needle_find_packed8_host: host rows of at most 256 chars -> (bitmap words, uint16[n]: see find_packed8_batch)."""
        L = _lib.lib()
        rows = np.ascontiguousarray(rows)
        if rows.dtype == np.int16:
            rows = rows.view(np.uint16)
        n, stride = rows.shape
        v = BatchView()
        v.rows, v.char_width, v.n_rows, v.row_stride, v.row_len = rows.ctypes.data, rows.dtype.itemsize, n, stride, stride
        if lengths is not None:
            lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
            v.lengths = lengths.ctypes.data
        words = np.zeros((n + 63) // 64, dtype=np.uint64)
        sl = np.zeros(n, dtype=np.uint16)
        _check(L.needle_find_packed8_host(self._h, ctypes.byref(v), words.ctypes.data, sl.ctypes.data))
        return words, sl

    MATCH_REC = np.dtype([("row", np.uint32), ("start", np.uint16), ("end", np.uint16)])  # needle_match_rec

    def find_compact(self, rows, lengths=None, stream=None, out=None):
        """find() with the MATCHED rows only, in row order, as {row u32, start u16, end u16} records
        (needle_find_compact_dev / _host: 8 bytes per matched row instead of 8 per row; rows of at most 65 534 chars).
        Host rows (numpy) -> (bitmap words, records as a MATCH_REC array); device rows (torch) -> (bitmap words,
        records int32[cap, 2] tensor whose rows are the raw records, n_matched as a 1-element int64 device tensor);
        out: optional caller-owned (bitmap int64, records int32[cap, 2], count int64[1]) device tensors."""
        L = _lib.lib()
        if isinstance(rows, np.ndarray):
            rows = np.ascontiguousarray(rows)
            if rows.dtype == np.int16:
                rows = rows.view(np.uint16)
            assert rows.ndim == 2 and rows.dtype in (np.uint8, np.uint16)
            n, stride = rows.shape
            v = BatchView()
            v.rows, v.char_width, v.n_rows, v.row_stride, v.row_len = rows.ctypes.data, rows.dtype.itemsize, n, stride, stride
            if lengths is not None:
                lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
                v.lengths = lengths.ctypes.data
            words = np.zeros((n + 63) // 64, dtype=np.uint64)
            recs = np.zeros(n, dtype=self.MATCH_REC)
            m = ctypes.c_uint64(0)
            _check(L.needle_find_compact_host(self._h, ctypes.byref(v), words.ctypes.data, recs.ctypes.data, n, ctypes.byref(m)))
            return words, recs[:m.value]
        import torch
        v = self._dev_view(rows, lengths)
        n = rows.shape[0]
        with torch.cuda.device(rows.device):
            s = torch.cuda.current_stream(rows.device).cuda_stream if stream is None else stream
            if out is not None:
                words, recs, cnt = out
            else:
                words = torch.empty((n + 63) // 64, dtype=torch.int64, device=rows.device)
                recs = torch.empty((n, 2), dtype=torch.int32, device=rows.device)
                cnt = torch.zeros(1, dtype=torch.int64, device=rows.device)
            _check(L.needle_find_compact_dev(self._h, ctypes.byref(v), words.data_ptr(), recs.data_ptr(), recs.shape[0], cnt.data_ptr(), s))
        return words, recs, cnt

    def find_packed16_host(self, rows, lengths=None):
        """needle_find_packed16_host: host rows -> (bitmap words, uint32[n]: low half start, high half end, 0xFFFF = no match)."""
        L = _lib.lib()
        rows = np.ascontiguousarray(rows)
        if rows.dtype == np.int16:
            rows = rows.view(np.uint16)
        n, stride = rows.shape
        v = BatchView()
        v.rows, v.char_width, v.n_rows, v.row_stride, v.row_len = rows.ctypes.data, rows.dtype.itemsize, n, stride, stride
        if lengths is not None:
            lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
            v.lengths = lengths.ctypes.data
        words = np.zeros((n + 63) // 64, dtype=np.uint64)
        se = np.zeros(n, dtype=np.uint32)
        _check(L.needle_find_packed16_host(self._h, ctypes.byref(v), words.ctypes.data, se.ctypes.data))
        return words, se

    # ---- haystacks packed back to back (one char buffer + offsets: what a JNI host gets from a String[])
    def _run_packed_host(self, op, data, offsets):
        L = _lib.lib()
        data = np.ascontiguousarray(data)
        if data.dtype == np.int16:
            data = data.view(np.uint16)
        assert data.ndim == 1 and data.dtype in (np.uint8, np.uint16), "data: 1-D uint8/uint16 code units"
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        v = _lib.PackedView()
        v.data, v.char_width, v.n_rows, v.offsets = data.ctypes.data, data.dtype.itemsize, n, offsets.ctypes.data
        words = np.zeros((n + 63) // 64, dtype=np.uint64)
        if op == "find":
            st = np.full(n, -1, dtype=np.int32)
            en = np.full(n, -1, dtype=np.int32)
            _check(L.needle_find_packed_host(self._h, ctypes.byref(v), words.ctypes.data, st.ctypes.data, en.ctypes.data))
            return words, st, en
        fn = L.needle_matches_packed_host if op == "matches" else L.needle_contained_in_packed_host
        _check(fn(self._h, ctypes.byref(v), words.ctypes.data))
        return words

    def matches_packed(self, data, offsets):
        return self._run_packed_host("matches", data, offsets)

    def contained_in_packed(self, data, offsets):
        return self._run_packed_host("contained_in", data, offsets)

    def find_packed(self, data, offsets):
        return self._run_packed_host("find", data, offsets)

    def find_strings(self, strings):
        """find() over a list of str (UTF-16 code units, like java.lang.String) -> (matched bool[n], start, end)."""
        data, offsets = pack_strings(strings)
        words, st, en = self.find_packed(data, offsets)
        return unpack_bitmap(words, len(strings)), st, en

    @staticmethod
    def rows_from_packed(data, offsets, row_stride=None, stream=None):
        """Device tensors: packed code units (1-D uint8 | int16) + int64 offsets[n + 1] -> (rows [n, row_stride],
        lengths int32[n]) in the fixed-stride layout the scan kernels read.  row_stride defaults to the longest row
        rounded up to 16 bytes (one device->host sync to learn it)."""
        import torch
        L = _lib.lib()
        assert data.is_cuda and data.dim() == 1 and data.is_contiguous() and data.dtype in (torch.uint8, torch.int16, torch.uint16)
        assert offsets.is_cuda and offsets.dtype == torch.int64 and offsets.dim() == 1 and offsets.is_contiguous()
        n = offsets.numel() - 1
        cw = data.element_size()
        per = 16 // cw
        if row_stride is None:
            longest = int((offsets[1:] - offsets[:-1]).max().item()) if n else 0
            row_stride = max(per, (longest + per - 1) // per * per)
        assert row_stride % per == 0
        with torch.cuda.device(data.device):
            s = torch.cuda.current_stream(data.device).cuda_stream if stream is None else stream
            rows = torch.empty((n, row_stride), dtype=data.dtype, device=data.device)
            lengths = torch.empty(n, dtype=torch.int32, device=data.device)
            overflow = torch.zeros(1, dtype=torch.int32, device=data.device)
            v = _lib.PackedView()
            v.data, v.char_width, v.n_rows, v.offsets = data.data_ptr(), cw, n, offsets.data_ptr()
            _check(L.needle_rows_from_packed_dev(ctypes.byref(v), rows.data_ptr(), row_stride, lengths.data_ptr(),
                                                 overflow.data_ptr(), s))
        return rows, lengths, overflow


def pack_strings(strings):
    """list[str] -> (uint16 code units back to back, uint64 offsets[n + 1])."""
    units = [_utf16(x) for x in strings]
    offsets = np.zeros(len(units) + 1, dtype=np.uint64)
    if units:
        offsets[1:] = np.cumsum([u.size for u in units])
    data = np.concatenate(units) if units else np.zeros(0, dtype=np.uint16)
    return data.astype(np.uint16), offsets


def unpack_bitmap(words, n_rows):
    """bitmap words -> bool array of n_rows (numpy)."""
    if not isinstance(words, np.ndarray):
        words = words.cpu().numpy()
    w = np.ascontiguousarray(words).view(np.uint64)
    bits = np.unpackbits(w.view(np.uint8), bitorder="little")
    return bits[:n_rows].astype(bool)


class DFACompiler:
    """DFACompiler.compile / compileToBytes entry point (DFACompiler.java:16-74): regex -> Pattern."""

    @staticmethod
    def compile(regex, class_name=None, flags=0):
        if regex is None:
            raise TypeError("regex string cannot be null")  # Objects.requireNonNull, RegexParser.java:87
        if flags & ~ALL_FLAGS:
            raise ValueError("Unknown flag bits")  # CompilerOptions.java:9-16
        u = _utf16(regex)
        h = ctypes.c_void_p()
        _check(_lib.lib().needle_compile(u.ctypes.data, u.size, int(flags), ctypes.byref(h)))
        return Pattern(h, regex, flags)
