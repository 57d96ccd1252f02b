#!/usr/bin/env python3
"""Generates tests/golden/snapshots/*.json by EXECUTING the reference's own compiled matchers.

Runs only in the build container (needs /root/reference).  For each of the 12 generated classes in
/root/reference/needle-compiler/src/test/resources/snapshots/ (regex list:
needle-compiler/src/test/java/com/justinblank/strings/SnapshotTests.java:30-57, compiled with flags 0)
it records

  * the tables the class carries (char->class map as runs, row stride N, the four STATES_* tables in the
    reference's own string encoding, accepting sets, per-method maxChar constants, how find() derives
    start) -- DATA extracted from static initialisers / constants, no reference source text;
  * inputs -> outputs of matches() / containedIn() / find()+start()+end() (and a second find()) on
    seeded haystacks, computed by interpreting the class's bytecode (oracle/jvm_snapshot.py).

Usage:  python tests/golden/gen_snapshot_vectors.py
"""
import json
import os
import random
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.jvm_snapshot import Machine, SnapshotMatcher, JavaThrow, instruction_starts  # noqa: E402

SNAP = "/root/reference/needle-compiler/src/test/resources/snapshots/"
OUT = os.path.join(ROOT, "tests", "golden", "snapshots")

# (class name, regex) -- SnapshotTests.java:30-57
EXAMPLES = [
    ("Sherlock", "Sherlock"),
    ("SherlockStreet", "Sherlock|Street"),
    ("SherlockInitialCharCaseInsensitive", "[Ss]herlock"),
    ("UnionOfManyNames", "Sherlock|Holmes|Watson|Irene|Adler|John|Baker"),
    ("Suffix", "anywhere|somewhere"),
    ("HolmesNearWatson", "Holmes.{1,10}Watson|Watson.{1,10}Holmes"),
    ("TwoNamesCaseInsensitiveFirstChar", "([Ss]herlock)|([Hh]olmes)"),
    ("aDotc", "a.c"),
    ("DigitPlus", "[0-9]+"),
    ("SingleCharacterUnicode", "ε"),
    ("UnicodeUnion", "ε|λ"),
    ("RepeatingUnionOfShortStrings", "(ab|a|bcdef|g)+"),
]

# strings that exercise each regex (full matches, near misses) -- written for this generator
SEEDS = {
    "Sherlock": ["Sherlock", "Sherloc", "SSherlock", "Sherlockk", "sherlock", "SherlocSherlock"],
    "SherlockStreet": ["Sherlock", "Street", "Stree", "SStreet", "SherStreet", "StreetSherlock", "Sherlockt"],
    "SherlockInitialCharCaseInsensitive": ["Sherlock", "sherlock", "SHerlock", "ssherlock", "herlock", "Ssherlock"],
    "UnionOfManyNames": ["Sherlock", "Holmes", "Watson", "Irene", "Adler", "John", "Baker", "Joh", "IreneAdler",
                         "HolmesWatson", "Bake", "AdleJohn", "WatsoWatson"],
    "Suffix": ["anywhere", "somewhere", "where", "nowhere", "anywher", "somewhereanywhere", "anysomewhere"],
    "HolmesNearWatson": ["Holmes Watson", "Watson Holmes", "HolmesWatson", "Holmes and Dr Watson", "Holmes1234567890Watson",
                         "Holmes12345678901Watson", "WatsonxHolmes", "Holmes\nWatson", "Watson, Holmes", "HolmesxWatsonxHolmes",
                         "Holmes Holmes Watson", "WatsonWatson Holmes"],
    "TwoNamesCaseInsensitiveFirstChar": ["Sherlock", "sherlock", "Holmes", "holmes", "HOlmes", "sherlocHolmes", "hholmes",
                                         "SherlockHolmes", "olmes"],
    "aDotc": ["abc", "a.c", "a\nc", "a\rc", "ac", "aac", "aεc", "a￿c", "abcabc", "ab", "xa c", "aaac"],
    "DigitPlus": ["0", "9", "12345", "a1", "1a", "ab12cd345", "/", ":", "a", "١", "12ε34", "x￿7"],
    "SingleCharacterUnicode": ["ε", "εε", "aε", "εa", "δ", "ζ", "e", "￿ε"],
    "UnicodeUnion": ["ε", "λ", "ελ", "aλ", "λa", "κ", "μ", "l", "￿λ"],
    "RepeatingUnionOfShortStrings": ["ab", "a", "bcdef", "g", "abab", "abcdef", "abcde", "gab", "bcdefg", "aabcdefg",
                                     "bcde", "xgx", "abg", "babcdef", "ababcdefgg"],
}


def _push_const(cf, code, pc):
    """If code[pc] pushes an int constant return (value, next_pc) else None."""
    op = code[pc]
    if 0x02 <= op <= 0x08:
        return op - 3, pc + 1
    if op == 0x10:
        v = code[pc + 1]
        return (v - 256 if v > 127 else v), pc + 2
    if op == 0x11:
        return struct.unpack(">h", code[pc + 1:pc + 3])[0], pc + 3
    if op == 0x12:
        return cf.const(code[pc + 1]), pc + 2
    if op == 0x13:
        return cf.const(struct.unpack(">H", code[pc + 1:pc + 3])[0]), pc + 3
    return None


def _iload(code, pc):
    op = code[pc]
    if op == 0x15:
        return code[pc + 1], pc + 2
    if 0x1A <= op <= 0x1D:
        return op - 0x1A, pc + 1
    return None


def _istore(code, pc):
    op = code[pc]
    if op == 0x36:
        return code[pc + 1], pc + 2
    if 0x3B <= op <= 0x3E:
        return op - 0x3B, pc + 1
    return None


def find_max_char(cf, name, desc):
    """Scan for `charAt; istore v; iload v; push K; if_icmple` -> K (the generated maxChar check)."""
    m = cf.methods.get((name, desc))
    if m is None:
        return None
    code = m["code"]["code"]
    for pc in instruction_starts(code):
        if code[pc] != 0xB6:  # invokevirtual
            continue
        if cf.member(struct.unpack(">H", code[pc + 1:pc + 3])[0])[1] != "charAt":
            continue
        s = _istore(code, pc + 3)
        if not s:
            continue
        l = _iload(code, s[1])
        if not l or l[0] != s[0]:
            continue
        c = _push_const(cf, code, l[1])
        if not c:
            continue
        if code[c[1]] == 0xA4:  # if_icmple
            return c[0]
    return None


def find_stride(cf):
    """Row stride N: `push N; imul` in matches()."""
    code = cf.methods[("matches", "()Z")]["code"]["code"]
    for pc in instruction_starts(code):
        c = _push_const(cf, code, pc)
        if c and code[c[1]] == 0x68:
            return c[0]
    return 1  # N == 1 would be folded; not seen


def find_fixed_len(cf):
    """find(II): `iload_3; push K; isub; putfield start` -> K (Factorization.canOnlyHaveOneLength)."""
    code = cf.methods[("find", "(II)Z")]["code"]["code"]
    for pc in instruction_starts(code):
        c = _push_const(cf, code, pc)
        if c and code[c[1]] == 0x64 and code[c[1] + 1] == 0xB5:
            if cf.member(struct.unpack(">H", code[c[1] + 2:c[1] + 4])[0])[1] == "start":
                return c[0]
    return None


def backwards_kind(cf):
    if ("indexBackwards", "(II)I") not in cf.methods:
        return {"kind": "fixed_len", "len": find_fixed_len(cf)}
    code = cf.methods[("indexBackwards", "(II)I")]["code"]["code"]
    uses_table = False
    for pc in instruction_starts(code):
        if code[pc] == 0xB2 and cf.member(struct.unpack(">H", code[pc + 1:pc + 3])[0])[1] == "STATES_BACKWARDS":
            uses_table = True
    if uses_table:
        return {"kind": "dfa"}
    # single-character reverse scan (DFAClassBuilder.generateSingleCharacterReverseScan): `push c; ...charAt; if_icmpne`
    for pc in instruction_starts(code):
        c = _push_const(cf, code, pc)
        if c and c[1] < len(code) and code[c[1]] == 0x19:  # aload string follows the pushed char
            return {"kind": "single_char_scan", "char": c[0]}
    raise RuntimeError("unrecognised indexBackwards")


def runs(arr):
    out = []
    start = 0
    for i in range(1, len(arr) + 1):
        if i == len(arr) or arr[i] != arr[start]:
            out.append([start, i - 1, arr[start]])
            start = i
    return out


def extract_tables(m):
    cf = m.cf
    n = find_stride(cf)
    bc = m.statics["BYTE_CLASSES"]
    out = {"stride": n, "class_map_runs": runs(bc[:65536]), "class_map_catch_all": bc[65536], "dfas": {}}
    strings = {f["name"]: f["const"] for f in cf.fields if f["name"].startswith("BYTE_CLASS_STRING_")}
    for key, meth, desc in (("Matches", "matches", "()Z"), ("ContainedIn", "containedIn", "()Z"),
                            ("Forwards", "indexForwards", "(II)I"), ("Backwards", "indexBackwards", "(II)I")):
        arr = m.statics["STATES_" + key.upper()]
        nst = len(arr) // n
        assert nst * n == len(arr)
        elem = [f["desc"] for f in cf.fields if f["name"] == "STATES_" + key.upper()][0]
        acc = [s for s in range(nst) if m.invoke("wasAccepted" + key, "(I)Z", [None, s])]
        tstr = [strings[k] for k in sorted(strings) if k.startswith("BYTE_CLASS_STRING_STATES_" + key.upper())]
        out["dfas"][key] = {
            "n_states": nst,
            "elem": "int16" if elem == "[S" else "int8",
            "table_strings": tstr,
            "table": list(arr),
            "accepting": acc,
            "accepts_dead": bool(m.invoke("wasAccepted" + key, "(I)Z", [None, -1])),
            "max_char": find_max_char(cf, meth, desc),
        }
    out["backwards"] = backwards_kind(cf)
    return out


def haystacks(name, regex, rng):
    alpha = sorted(set(c for c in regex if c.isalnum() or ord(c) > 127))
    noise = list("xyz qQ.,\n\r19") + ["ε", "λ", "é", "￿", "中"]
    pool = alpha * 3 + noise
    out = [""]
    out += SEEDS[name]
    # seeds embedded in noise
    for s in SEEDS[name]:
        for _ in range(4):
            pre = "".join(rng.choice(pool) for _ in range(rng.randrange(0, 12)))
            post = "".join(rng.choice(pool) for _ in range(rng.randrange(0, 12)))
            out.append(pre + s + post)
    # pure noise, short and long
    for _ in range(150):
        ln = rng.choice([1, 2, 3, 5, 8, 13, 21, 34, 64])
        out.append("".join(rng.choice(pool) for _ in range(ln)))
    # words of the regex alphabet only (dense near-misses)
    for _ in range(100):
        ln = rng.randrange(1, 40)
        out.append("".join(rng.choice(alpha) for _ in range(ln)))
    # two seeds in one haystack (second find())
    for _ in range(40):
        a, b = rng.choice(SEEDS[name]), rng.choice(SEEDS[name])
        mid = "".join(rng.choice(pool) for _ in range(rng.randrange(0, 20)))
        out.append(a + mid + b)
    seen, uniq = set(), []
    for h in out:
        if h not in seen:
            seen.add(h)
            uniq.append(h)
    return uniq


def run_one(m, h):
    rec = {"h": h}
    try:
        rec["matches"] = SnapshotMatcher(m, h).matches()
    except JavaThrow as e:
        rec["matches"] = "throw:" + str(e).split(" ")[0]
    try:
        rec["containedIn"] = SnapshotMatcher(m, h).containedIn()
    except JavaThrow as e:
        rec["containedIn"] = "throw:" + str(e).split(" ")[0]
    try:
        sm = SnapshotMatcher(m, h)
        f1 = sm.find()
        rec["find"] = [f1, sm.start(), sm.end()]
        # repeated find(): nextStart cursor semantics (DFAClassBuilder.java:616-659); bounded against
        # the reference's empty-match non-advance
        if f1 and sm.end() > sm.start():
            f2 = sm.find()
            rec["find2"] = [f2, sm.start(), sm.end()]
    except JavaThrow as e:
        rec["find"] = "throw:" + str(e).split(" ")[0]
    return rec


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, regex in EXAMPLES:
        rng = random.Random(0x5EED0000 + sum(map(ord, name)))
        m = Machine(SNAP + name + ".class")
        doc = {"name": name, "regex": regex, "flags": 0,
               "source": "interpreted bytecode of needle-compiler/src/test/resources/snapshots/%s.class" % name}
        doc.update(extract_tables(m))
        doc["vectors"] = [run_one(m, h) for h in haystacks(name, regex, rng)]
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(doc, f, ensure_ascii=True, separators=(",", ":"))
        nthrow = sum(1 for v in doc["vectors"] if any(isinstance(v.get(k), str) for k in ("matches", "containedIn", "find")))
        print(name, "states", {k: d["n_states"] for k, d in doc["dfas"].items()}, "stride", doc["stride"],
              "maxchar", {k: d["max_char"] for k, d in doc["dfas"].items()}, doc["backwards"],
              "vectors", len(doc["vectors"]), "throws", nthrow)


if __name__ == "__main__":
    main()
