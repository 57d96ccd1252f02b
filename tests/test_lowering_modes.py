"""How automata are lowered for the device (needle_pattern_program_info: host-side, no GPU): the mode ladder of
needle_amd/csrc/needle_lower.cpp -- packed functions, pair table, LDS tables, and for tables larger than the LDS the
compressed whole-automaton form (dense rows + exception records, verified cell by cell against the dense table inside
lower()) before the hot-rows and HBM-table fallbacks -- plus window addressing.  The reference fixes the corresponding
choices per generated class (useShorts, getEffectiveByteClassCount: DFAClassBuilder.java:79-85, :240-253)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, json
sys.path.insert(0, ".")
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler
lo, hi, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p = DFACompiler.compile("|".join(W.keywords(n, min_len=lo, max_len=hi)), "t", 0)
print(json.dumps({w: p.program_info(w, cw) for w in ("forwards", "contained_in", "matches") for cw in (1,)}))
'''


def _info(lo, hi, n, **env):
    import json
    r = subprocess.run([sys.executable, "-c", CODE, str(lo), str(hi), str(n)], env=dict(os.environ, **env), capture_output=True, text=True,
                       cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_sparse_match_dictionary_is_lds_resident_as_compressed_automaton():
    """C3-sparse (1000 keywords of 6..8 chars: 4439 states, a 275 KB dense table): the whole automaton fits the LDS next to
    16 waves x 64-byte tiles as ~600 dense rows + ~4000 records (some chained), window addressing over [a-z]."""
    i = _info(6, 8, 1000)
    for w in ("forwards", "contained_in", "matches"):
        d = i[w]
        assert d["mode"] == 6 and d["waves"] == 16 and d["tile_bytes"] == 64, (w, d)
        assert d["lds_bytes"] + 16 * 64 * 64 <= 160 * 1024 and d["blob_bytes"] == d["lds_bytes"]
        assert 400 <= d["dense_rows"] <= 900 and 2000 <= d["records"] <= 5000
        assert d["window"] == 1 and d["window_lo"] == ord("a") - 1 and d["window_hi"] == ord("z") + 1
    assert i["forwards"]["chains"] == 1 and i["forwards"]["lengths_form"] == 1


def test_fallback_ladder():
    off = _info(6, 8, 1000, NEEDLE_SPARSE="0")
    assert off["forwards"]["mode"] == 5 and off["forwards"]["hot_rows"] > 1000 and off["forwards"]["window"] == 1
    assert _info(6, 8, 1000, NEEDLE_SPARSE="0", NEEDLE_HYBRID="0")["forwards"]["mode"] == 3
    cmap = _info(6, 8, 1000, NEEDLE_WINDOW="0")
    assert cmap["forwards"]["mode"] == 6 and cmap["forwards"]["window"] == 0
    # 2000 keywords of 5..9 chars (8505 states): more records than the 16-bit record addresses reach -> hot rows
    assert _info(5, 9, 2000)["forwards"]["mode"] == 5


def test_dictionary_that_fits_keeps_its_dense_table():
    i = _info(3, 5, 1000)
    assert i["forwards"]["mode"] == 2 and i["forwards"]["window"] == 1 and i["forwards"]["waves"] == 16
    assert i["matches"]["mode"] == 2


@pytest.mark.parametrize("regex,mode", [("[0-9]+", 0), ("http://.+", 4), ("Sherlock|Holmes|Watson|Irene|Adler|John|Baker", 4)])
def test_small_automata(regex, mode):
    from needle_amd.pattern import DFACompiler
    p = DFACompiler.compile(regex, "t", 0)
    d = p.program_info("forwards")
    assert d["mode"] == mode and d["window"] == 0 and d["tile_bytes"] == 128, d
    if mode == 0:  # packed functions: 64 KiB of F; find() adds the packed backward automaton and gives up ONE wave for it
        assert d["waves"] == 15 and p.program_info("contained_in")["waves"] == 16
    else:
        assert d["waves"] >= 12


def test_find_takes_the_lengths_form_on_lds_tables():
    """find() on an LDS-table automaton whose pattern allows it (bounded match length, not nullable) is lowered as the
    "lengths" automaton (needle_lower.h: start = end - the length the stop state remembers; DFAClassBuilder.java:640-656
    generalised per state) -- no backward program; NEEDLE_FIND_LENGTHS=0 brings the indexBackwards form back; the
    compressed form (mode 6) and unbounded patterns keep it."""
    i = _info(3, 5, 1000)
    assert i["forwards"]["mode"] == 2 and i["forwards"]["lengths_form"] == 1
    assert i["contained_in"]["lengths_form"] == 0
    off = _info(3, 5, 1000, NEEDLE_FIND_LENGTHS="0")
    assert off["forwards"]["mode"] == 2 and off["forwards"]["lengths_form"] == 0
    assert off["forwards"]["n_states"] <= i["forwards"]["n_states"]
    # the compressed form carries the lengths automaton too (4487 states; END records instead of an END column): round 3
    big = _info(6, 8, 1000)["forwards"]
    assert big["mode"] == 6 and big["lengths_form"] == 1 and big["lds_bytes"] + 16 * 64 * 64 <= 160 * 1024
    assert _info(6, 8, 1000, NEEDLE_FIND_LENGTHS_SPARSE="0")["forwards"]["lengths_form"] == 0
    from needle_amd.pattern import DFACompiler
    names = DFACompiler.compile("Sherlock|Holmes|Watson|Irene|Adler|John|Baker", "t", 0).program_info("forwards", 1)
    assert names["mode"] == 4 and names["lengths_form"] == 1  # the pair table carries the lengths automaton too (40 states)
    assert DFACompiler.compile("[0-9]+x", "t", 0).program_info("forwards", 1)["lengths_form"] == 0  # unbounded
    assert DFACompiler.compile("(ab|a|bcdef|g)x", "t", 0).program_info("forwards", 1)["lengths_form"] in (0, 1)


def test_c5w_is_a_multi_class_utf16_table_automaton(oracle_lib):
    """C5's wide variant (needle_amd/workload.py SEQ_ALTS): tens of classes, tens of states; on UTF-16 rows a plain
    uint8 LDS table behind the two-level page map (no pair table: 8-bit rows only).  The generator's planted instances match, and
    the oracle agrees with Python's `re` on the generated rows."""
    import re
    import numpy as np
    from needle_amd import workload as W
    from test_compile_matches_txt import oracle_for
    from needle_amd.pattern import DFACompiler
    rx = W.scriptseq_regex()
    p = DFACompiler.compile(rx, "ScriptSeq", 0)
    inf = p.info()
    assert inf["stride"] >= 20 and inf["n_states"]["forwards"] > 20 and inf["n_states"]["contained_in"] > 20, inf
    for w in ("forwards", "contained_in", "matches"):
        d = p.program_info(w, 2)
        # (the flat 64 KB page map -- one column lookup per char -- next to 16 waves x 64-byte tiles; NEEDLE_FLAT_MAP=0: the compact
        # two-level map and 128-byte tiles)
        assert d["mode"] == 1 and d["waves"] == 16 and d["tile_bytes"] == 64 and 66048 < d["lds_bytes"] < 80000, (w, d)
    cre = re.compile(rx)
    for w in W.scriptseq_instances():
        assert cre.fullmatch("".join(map(chr, w))), w
    o, _ = oracle_for(rx, 0)
    rows = W.scriptseq_batch(np, 3, 1500, 256)
    m, s, e = o.batch_find(rows, threads=4)
    assert 0.25 < m.mean() < 0.6
    for i in range(len(rows)):
        mm = cre.search("".join(map(chr, rows[i])))
        assert ((True, mm.start(), mm.end()) if mm else (False, -1, -1)) == (bool(m[i]), int(s[i]), int(e[i])), i
