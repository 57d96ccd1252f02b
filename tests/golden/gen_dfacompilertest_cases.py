#!/usr/bin/env python3
"""Transcribes the inline known-answer asserts of the reference's
needle-compiler/src/test/java/com/justinblank/strings/DFACompilerTest.java (lines 43-660: one regex per @Test, then
match(...) / fail(...) / find(...) helper calls and assertTrue/False(pattern.matcher(s).matches()/containedIn()))
into tests/golden/dfacompilertest_cases.json as neutral (regex, flags, op, string) records.  The helpers' meaning is
SearchMethodTestUtil.java:48-120 and is re-implemented in tests/test_reference_asserts.py.  Runs only in the build
container; no reference source text is copied, only the literals of the asserts."""
import json
import os
import re

SRC = "/root/reference/needle-compiler/src/test/java/com/justinblank/strings/DFACompilerTest.java"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dfacompilertest_cases.json")
FLAGS = {"CASE_INSENSITIVE": 0x02, "DOTALL": 0x20, "UNICODE_CASE": 0x40, "UNICODE_CHARACTER_CLASS": 0x100,
         "LEFTMOST_LONGEST": 0x800000}
STR = r'"((?:[^"\\]|\\.)*)"'


def unescape(s):
    out, i = [], 0
    while i < len(s):
        c = s[i]
        if c != "\\":
            out.append(c); i += 1; continue
        n = s[i + 1]
        if n == "u":
            out.append(chr(int(s[i + 2:i + 6], 16))); i += 6
        else:
            out.append({"n": "\n", "r": "\r", "t": "\t", "\\": "\\", '"': '"', "'": "'", "0": "\0", "f": "\f", "b": "\b"}[n]); i += 2
    return "".join(out)


def flags_of(expr):
    v = 0
    for name, bit in FLAGS.items():
        if re.search(r"\b%s\b" % name, expr):
            v |= bit
    return v


def main():
    src = open(SRC, encoding="utf-8").read()
    tests = re.split(r"@Test", src)[1:]
    cases = []
    for body in tests:
        m = re.search(r"void\s+(\w+)\s*\(", body)
        name = m.group(1)
        if name in ("fileBasedTests",):
            continue
        pats = {}  # variable -> (regex, flags)
        ops = []
        for line in body.split("\n"):
            line = line.strip()
            c = re.search(r"(\w+)\s*=\s*(?:DFACompiler\.compile|anonymousPattern)\(\s*" + STR + r"\s*(?:,\s*" + STR + r")?\s*(?:,\s*([^;]*))?\)\s*;", line)
            if c and "+" not in line.split("=", 1)[1].split('"')[0]:
                var, rx, _cls, rest = c.group(1), unescape(c.group(2)), c.group(3), c.group(4) or ""
                if re.search(r'"\s*\+', line) or re.search(r'\+\s*"', line.split("compile(")[-1].split(",")[0] if "compile(" in line else ""):
                    continue
                pats[var] = (rx, flags_of(rest))
                continue
            for op in ("match", "fail", "find"):
                mm = re.match(r"(?:return\s+)?" + op + r"\(\s*(\w+)\s*,\s*" + STR + r"\s*(?:,\s*([^)]*))?\)\s*;", line)
                if mm and mm.group(1) in pats:
                    rx, fl = pats[mm.group(1)]
                    extra = (mm.group(3) or "").strip()
                    rec = {"test": name, "regex": rx, "flags": fl, "op": op, "s": unescape(mm.group(2))}
                    if op == "find" and extra:
                        if re.fullmatch(r"\d+\s*,\s*\d+", extra):
                            a, b = [int(x) for x in extra.split(",")]
                            rec["range"] = [a, b]
                        else:
                            rec["op"] = "find_noise"  # find(pattern, needle, prefix, suffix) under QuickTheory
                    ops.append(rec)
            mm = re.match(r"assert(True|False)\(\s*(\w+)\.matcher\(\s*" + STR + r"\s*\)\.(matches|containedIn)\(\)\s*(?:,.*)?\)\s*;", line)
            if mm and mm.group(2) in pats:
                rx, fl = pats[mm.group(2)]
                ops.append({"test": name, "regex": rx, "flags": fl, "op": "assert_" + mm.group(4), "s": unescape(mm.group(3)),
                            "expect": mm.group(1) == "True"})
        cases.extend(ops)
    json.dump({"source": "DFACompilerTest.java inline asserts", "cases": cases}, open(OUT, "w"), ensure_ascii=True, indent=0)
    by = {}
    for c in cases:
        by[c["op"]] = by.get(c["op"], 0) + 1
    print(len(cases), "cases from", len(set(c["test"] for c in cases)), "tests", by, "regexes", len(set((c["regex"], c["flags"]) for c in cases)))


if __name__ == "__main__":
    main()
