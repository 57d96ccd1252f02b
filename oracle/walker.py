"""ctypes front end of oracle/needle_walk.c (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Also holds the table decoding the generated classes do in their static initialisers:
  * decode_table_strings  <- ByteClassUtil.fillMultipleByteClassesFromString[UsingShorts]_singleArray
                             needle-types/src/main/java/com/justinblank/strings/ByteClassUtil.java:50-120
                             over an array pre-filled with -1 (DFAClassBuilder.populateByteClassArrays,
                             needle-compiler/.../DFAClassBuilder.java:317-333)
  * class_map_from_runs   <- the Arrays.fill(0) + ByteClassUtil.fillBytes run list emitted by
                             DFAClassBuilder.addByteClasses, DFAClassBuilder.java:269-305
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libneedle_oracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "needle_walk.c")):
            build()
        L = ctypes.CDLL(path)
        P, I32, I64, U32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32
        L.ndl_matches.argtypes = [P, P, ctypes.c_int, I32]
        L.ndl_matches.restype = ctypes.c_int
        L.ndl_contained_in.argtypes = [P, P, ctypes.c_int, I32]
        L.ndl_contained_in.restype = ctypes.c_int
        L.ndl_index_forwards.argtypes = [P, P, ctypes.c_int, I32, I32]
        L.ndl_index_forwards.restype = I32
        L.ndl_index_backwards.argtypes = [P, P, ctypes.c_int, I32, I32]
        L.ndl_index_backwards.restype = I32
        L.ndl_find.argtypes = [P, P, I32, I32, P, ctypes.c_int, I32, I32, P, P]
        L.ndl_find.restype = ctypes.c_int
        L.ndl_batch_matches.argtypes = [P, P, ctypes.c_int, I64, I64, P, U32, P, ctypes.c_int]
        L.ndl_batch_contained_in.argtypes = [P, P, ctypes.c_int, I64, I64, P, U32, P, ctypes.c_int]
        L.ndl_batch_find.argtypes = [P, P, I32, I32, P, ctypes.c_int, I64, I64, P, U32, P, P, P, ctypes.c_int]
        _LIB = L
    return _LIB


class _CDfa(ctypes.Structure):
    _fields_ = [("class_map", ctypes.c_void_p), ("stride", ctypes.c_int32), ("table", ctypes.c_void_p),
                ("accepting", ctypes.c_void_p), ("n_states", ctypes.c_int32), ("max_char", ctypes.c_int32)]


def decode_table_strings(strings, n_states, stride):
    """-> int16 array [n_states * stride], -1 where the strings carry no entry."""
    t = np.full(n_states * stride, -1, dtype=np.int16)
    for s in strings:
        for state_string in s.split(";"):
            if not state_string:
                continue
            st, trans = state_string.split(":")
            st = int(st, 16)
            for comp in trans.split(","):
                bc, tgt = comp.split("-")
                t[st * stride + int(bc, 16)] = int(tgt, 16)
    return t


def class_map_from_runs(runs):
    m = np.zeros(65536, dtype=np.uint8)
    for a, b, v in runs:
        m[a:b + 1] = v
    return m


class Dfa:
    """One of the four per-regex automata, in the generated class's own layout."""

    def __init__(self, class_map, stride, table, accepting, max_char):
        self.class_map = np.ascontiguousarray(class_map, dtype=np.uint8)
        assert self.class_map.size == 65536
        self.stride = int(stride)
        self.table = np.ascontiguousarray(table, dtype=np.int16)
        self.n_states = self.table.size // self.stride
        acc = np.zeros(self.n_states, dtype=np.uint8)
        for s in accepting:  # iterable of accepting state numbers
            acc[int(s)] = 1
        self.accepting = acc
        self.max_char = 0xFFFF if max_char is None else int(max_char)
        self._c = _CDfa(self.class_map.ctypes.data, self.stride, self.table.ctypes.data, self.accepting.ctypes.data,
                        self.n_states, self.max_char)

    @property
    def ptr(self):
        return ctypes.addressof(self._c)


def _chars(h):
    """str | bytes | ndarray -> (contiguous array, char width, length)."""
    if isinstance(h, (bytes, bytearray)):
        a = np.frombuffer(bytes(h), dtype=np.uint8)
        return a, 1, a.size
    if isinstance(h, str):
        a = np.array([ord(c) for c in h], dtype=np.uint16)
        return a, 2, a.size
    a = np.ascontiguousarray(h)
    assert a.dtype in (np.uint8, np.uint16)
    return a, a.dtype.itemsize, a.size


class OraclePattern:
    """The reference Matcher contract over oracle tables (single haystack + batch)."""

    def __init__(self, matches, contained_in, forwards, backwards, fixed_len=-1, single_char=-1):
        self.m, self.c, self.f, self.b = matches, contained_in, forwards, backwards
        self.fixed_len = -1 if fixed_len is None else int(fixed_len)
        self.single_char = -1 if single_char is None else int(single_char)

    @classmethod
    def from_fixture(cls, doc, backwards_as_dfa=False):
        cm = class_map_from_runs(doc["class_map_runs"])
        d = {}
        for key, spec in doc["dfas"].items():
            table = decode_table_strings(spec["table_strings"], spec["n_states"], doc["stride"])
            d[key] = Dfa(cm, doc["stride"], table, set(spec["accepting"]), spec["max_char"])
        bk = doc["backwards"]
        fixed = bk["len"] if bk["kind"] == "fixed_len" else -1
        single = bk["char"] if bk["kind"] == "single_char_scan" and not backwards_as_dfa else -1
        if backwards_as_dfa:
            fixed = -1
        return cls(d["Matches"], d["ContainedIn"], d["Forwards"], d["Backwards"], fixed, single)

    def matches(self, h):
        a, w, n = _chars(h)
        return bool(lib().ndl_matches(self.m.ptr, a.ctypes.data, w, n))

    def contained_in(self, h):
        a, w, n = _chars(h)
        return bool(lib().ndl_contained_in(self.c.ptr, a.ctypes.data, w, n))

    def find(self, h, start=0):
        """-> (matched, start, end) of the first find() from `start`; (False, None, -1) if none."""
        a, w, n = _chars(h)
        s, e = ctypes.c_int32(-1), ctypes.c_int32(-1)
        r = lib().ndl_find(self.f.ptr, self.b.ptr, self.fixed_len, self.single_char, a.ctypes.data, w, n, start,
                           ctypes.byref(s), ctypes.byref(e))
        if r == -2:
            raise RuntimeError("reference would throw (ArrayIndexOutOfBounds)")
        return (True, s.value, e.value) if r == 1 else (False, None, e.value)

    def find_all(self, h, limit=100000):
        """Repeated find() on one Matcher (nextStart = end, DFAClassBuilder.java:616-659) -> [(start, end), ...].
        The enumeration ends where the reference's cursor stops advancing: an empty match is reported once and ends
        it; so does any match that does not end beyond the cursor it was searched from.  A nullable pattern searched
        from cursor == length reports end = the literal 0 of DFAClassBuilder.java:356 with start = length (end <
        start): the reference would cycle (0,len),(len,0),... for ever; that wrapped pseudo-match is dropped here."""
        out, cur = [], 0
        while len(out) < limit:
            found, s, e = self.find(h, start=cur)
            if not found or e < s:
                break
            out.append((s, e))
            if e == s or e <= cur:
                break
            cur = e
        return out

    # -- batches: rows is a 2-D uint8/uint16 array [n_rows, stride]
    def _batch_args(self, rows, lengths):
        rows = np.ascontiguousarray(rows)
        assert rows.ndim == 2 and rows.dtype in (np.uint8, np.uint16)
        lp = None
        if lengths is not None:
            lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
            lp = lengths.ctypes.data
        return rows, rows.dtype.itemsize, rows.shape[0], rows.shape[1], lengths, lp

    def batch_matches(self, rows, lengths=None, threads=1):
        rows, w, n, stride, lengths, lp = self._batch_args(rows, lengths)
        out = np.zeros(n, dtype=np.uint8)
        lib().ndl_batch_matches(self.m.ptr, rows.ctypes.data, w, n, stride, lp, stride, out.ctypes.data, threads)
        return out.astype(bool)

    def batch_contained_in(self, rows, lengths=None, threads=1):
        rows, w, n, stride, lengths, lp = self._batch_args(rows, lengths)
        out = np.zeros(n, dtype=np.uint8)
        lib().ndl_batch_contained_in(self.c.ptr, rows.ctypes.data, w, n, stride, lp, stride, out.ctypes.data, threads)
        return out.astype(bool)

    def batch_find(self, rows, lengths=None, threads=1):
        rows, w, n, stride, lengths, lp = self._batch_args(rows, lengths)
        m = np.zeros(n, dtype=np.uint8)
        s = np.zeros(n, dtype=np.int32)
        e = np.zeros(n, dtype=np.int32)
        lib().ndl_batch_find(self.f.ptr, self.b.ptr, self.fixed_len, self.single_char, rows.ctypes.data, w, n, stride,
                             lp, stride, m.ctypes.data, s.ctypes.data, e.ctypes.data, threads)
        return m.astype(bool), s, e
