"""CPU-side checks of the drop-in boundary: libneedle_hip.so loads and exports every symbol that
include/needle_hip.h declares; argument validation works without a GPU (no compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_snapshot


@pytest.fixture(scope="module")
def lib():
    from needle_amd import build
    build.build()
    from needle_amd import _lib
    return _lib.lib()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "needle_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(needle_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    from needle_amd import _lib
    assert sorted(_lib.EXPORTS) == names


def test_version_and_error_strings(lib):
    assert b"needle_hip" in lib.needle_version()
    assert isinstance(lib.needle_last_error(), bytes)


def test_from_tables_roundtrip_and_validation(lib):
    from needle_amd.pattern import Pattern
    from oracle.walker import class_map_from_runs, decode_table_strings
    doc = load_snapshot("UnionOfManyNames")
    cm = class_map_from_runs(doc["class_map_runs"])
    names = {"matches": "Matches", "contained_in": "ContainedIn", "forwards": "Forwards", "backwards": "Backwards"}
    dfas = {k: dict(n_states=doc["dfas"][v]["n_states"], max_char=doc["dfas"][v]["max_char"],
                    accepting=doc["dfas"][v]["accepting"], table_strings=doc["dfas"][v]["table_strings"])
            for k, v in names.items()}
    p = Pattern.from_tables(cm, doc["stride"], dfas, fixed_len=-1)
    t = p.tables()
    assert t["stride"] == doc["stride"]
    assert (t["class_map"] == cm).all()
    for k, v in names.items():
        want = decode_table_strings(doc["dfas"][v]["table_strings"], doc["dfas"][v]["n_states"], doc["stride"])
        assert (t["dfas"][k]["table"] == want).all()
        assert t["dfas"][k]["accepting"] == doc["dfas"][v]["accepting"]
    info = p.info()
    # 32 device states x 31^2 column pairs x 2 B = 61 KB: the two-chars-per-lookup table fits the LDS
    assert info["n_states"]["matches"] == 31 and info["kernel_mode"]["matches"] == 4
    # malformed table string / out-of-range target -> ValueError, not a crash
    bad = dict(dfas)
    bad["matches"] = dict(dfas["matches"], table_strings=["0:zz-1"])
    with pytest.raises(ValueError):
        Pattern.from_tables(cm, doc["stride"], bad)
    bad["matches"] = dict(dfas["matches"], table_strings=["0:1-7f"])
    with pytest.raises(ValueError):
        Pattern.from_tables(cm, doc["stride"], bad)


def test_small_automata_lower_to_nibble_mode(lib):
    from test_gpu_parity import pattern_from_fixture
    p = pattern_from_fixture(load_snapshot("DigitPlus"))
    assert set(p.info()["kernel_mode"].values()) == {0}


def test_precompiled_blob_roundtrip(lib):
    """needle_pattern_serialize / deserialize (the Precompile analogue): tables survive bit for bit; damaged blobs
    are rejected, not crashed on."""
    from needle_amd.pattern import DFACompiler, Pattern
    p = DFACompiler.compile("Sherlock|Holmes|Watson|Irene|Adler|John|Baker", "UnionOfManyNames", 0)
    blob = p.to_bytes()
    assert blob[:4] == b"NDLT" and len(blob) > 65536
    q = Pattern.from_bytes(blob)
    a, b = p.tables(), q.tables()
    assert a["stride"] == b["stride"] and a["fixed_len"] == b["fixed_len"] and (a["class_map"] == b["class_map"]).all()
    assert (a["min_len"], a["max_len"]) == (b["min_len"], b["max_len"]) == (4, 8)
    for k in a["dfas"]:
        assert (a["dfas"][k]["table"] == b["dfas"][k]["table"]).all()
        assert a["dfas"][k]["accepting"] == b["dfas"][k]["accepting"] and a["dfas"][k]["max_char"] == b["dfas"][k]["max_char"]
    assert q.to_bytes() == blob
    for bad in (blob[:100], blob[:-1], b"XXXX" + blob[4:], blob + b"\0", blob[:8] + b"\xff\xff\xff\x7f" + blob[12:]):
        with pytest.raises(ValueError):
            Pattern.from_bytes(bad)


def test_packed_view_validation_without_a_gpu(lib):
    """Argument checks of the packed (offsets) entry points happen before any device call."""
    from needle_amd import _lib
    from needle_amd.pattern import DFACompiler
    p = DFACompiler.compile("[0-9]+", "d")
    data = np.frombuffer(b"ab12cd", dtype=np.uint8).copy()
    bad_offsets = np.array([0, 4, 2], dtype=np.uint64)  # decreasing
    with pytest.raises(ValueError):
        p.find_packed(data, bad_offsets)
    v = _lib.PackedView()
    v.data, v.char_width, v.n_rows, v.offsets = data.ctypes.data, 3, 1, bad_offsets.ctypes.data
    assert lib.needle_rows_from_packed_dev(ctypes.byref(v), None, 16, None, None, None) == _lib.ERR_INVALID
    v.char_width = 1
    assert lib.needle_rows_from_packed_dev(ctypes.byref(v), None, 16, None, None, None) == _lib.ERR_INVALID  # NULL outputs
    # zero rows: nothing to do, no device needed
    words = p.contained_in_packed(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.uint64))
    assert words.size == 0


def test_table_string_codec_known_answers(lib):
    """ByteClassUtilTest.java:17-58: the hex "state:class-target,...;..." codec (fillMultipleByteClassesFromString
    [UsingShorts]_singleArray), through the oracle's decoder and through needle_pattern_from_tables' table_string."""
    from needle_amd.pattern import Pattern
    from oracle.walker import decode_table_strings
    text = "0:1-2,2-3,3-c;1:1-3,2-4,3-13;2:1-4,2-5,3-e;c:1-d,2-e,3-17"
    want = {1: 2, 2: 3, 3: 12, 5: 3, 49: 13}
    t = decode_table_strings([text], 24, 4)
    for i, v in want.items():
        assert t[i] == v
    for i in range(4096):  # encodeDecode: hex round trip
        assert int(format(i, "x"), 16) == i
    cm = np.zeros(65536, dtype=np.uint8)
    spec = dict(n_states=24, max_char=0xFFFF, accepting=[23], table_strings=[text])
    p = Pattern.from_tables(cm, 4, {k: spec for k in ("matches", "contained_in", "forwards", "backwards")})
    got = p.tables()["dfas"]["matches"]["table"]
    for i, v in want.items():
        assert got[i] == v
    assert (got == t).all()
    with pytest.raises(ValueError):  # fill_bytes_from_string_errors_on_bad_input: "1-z"
        Pattern.from_tables(cm, 4, {k: dict(spec, table_strings=["0:1-z"]) for k in ("matches", "contained_in", "forwards", "backwards")})


def test_tuning_info_lists_every_environment_switch(lib):
    """needle_tuning_info (include/needle_hip.h): every getenv("NEEDLE_...") in the library's sources is in the table, and the table
    names nothing the sources do not read; the same list stands in INTEGRATION.md."""
    import glob
    need = ctypes.c_size_t(0)
    lib.needle_tuning_info.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    assert lib.needle_tuning_info(None, 0, ctypes.byref(need)) == 0 and need.value > 1000
    buf = ctypes.create_string_buffer(need.value)
    assert lib.needle_tuning_info(buf, need.value, None) == 0
    lines = buf.value.decode().strip().split("\n")
    assert lines[0].split("\t") == ["name", "default", "current", "scope", "effect"]
    rows = [ln.split("\t") for ln in lines[1:]]
    assert all(len(r) == 5 and r[4] for r in rows)
    listed = {r[0] for r in rows}
    used = set()
    for f in glob.glob(os.path.join(ROOT, "needle_amd", "csrc", "*")):
        if f.endswith(("needle_tuning.cpp", "_probe.hip")):
            continue
        used |= set(re.findall(r'getenv\("(NEEDLE_[A-Z0-9_]+)"\)', open(f).read()))
    assert used == listed, (sorted(used - listed), sorted(listed - used))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in listed:
        assert "`%s`" % name in doc, name
    # a short buffer gets a truncated, terminated copy
    small = ctypes.create_string_buffer(16)
    assert lib.needle_tuning_info(small, 16, None) == 0 and len(small.value) == 15
