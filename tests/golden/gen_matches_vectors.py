#!/usr/bin/env python3
"""Transcribes the reference's golden-vector file needle-compiler/src/test/resources/matches.txt (consumed by
DFACompilerTest.fileBasedTests, DFACompilerTest.java:701-773) into tests/golden/matches.json: one record per
row {pattern, haystack, found, start, end, flags|null}.  Tokenisation follows RegexTestSpecParser.java:32-143
(space-separated columns, '...' quoting, \\n / \\r unescaped in the haystack column, hex flags column).
Also records the known answers asserted inline by DFACompilerTest.java for the BASELINE regexes (C1 `http://.+`,
`[0-9]+`, repeated find(), find(int,int) windows, dot/DOTALL).  Runs only in the build container."""
import json
import os

SRC = "/root/reference/needle-compiler/src/test/resources/matches.txt"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "matches.json")


def jtrim(s):
    """java.lang.String.trim(): strips code points <= U+0020 only (NOT Unicode whitespace)."""
    a, b = 0, len(s)
    while a < b and s[a] <= " ":
        a += 1
    while b > a and s[b - 1] <= " ":
        b -= 1
    return s[a:b]


class Chomper:
    def __init__(self, s):
        self.s, self.idx = s, 0

    def more(self):
        return len(self.s) > self.idx

    def chomp(self):
        s, start, seen, inq = self.s, self.idx, False, False
        while self.idx < len(s):
            c = s[self.idx]
            if c == " " and not inq:
                if seen:
                    return jtrim(s[start:self.idx])
            elif c == "'":
                if inq:
                    self.idx += 1
                    sub = jtrim(s[start:self.idx])
                    return sub[1:-1]
                inq = True
            else:
                seen = True
            self.idx += 1
        if not seen:
            raise ValueError("nothing to chomp in %r" % s)
        return jtrim(s[start:self.idx])


def main():
    rows = []
    for line in open(SRC, encoding="utf-8").read().split("\n"):
        if not line.strip() or line.startswith("#"):
            continue
        ch = Chomper(jtrim(line))
        pattern = ch.chomp()
        target = ch.chomp().replace("\\n", "\n").replace("\\r", "\r")
        ok = ch.chomp() == "y"
        start = end = -1
        if ok:
            start, end = int(ch.chomp()), int(ch.chomp())
        flags = int(ch.chomp(), 16) if ch.more() else None
        rows.append({"pattern": pattern, "haystack": target, "found": ok, "start": start, "end": end, "flags": flags})
    # inline known answers of DFACompilerTest.java (line numbers of the asserts)
    inline = [
        {"src": "DFACompilerTest.java:524-540", "pattern": "http://.+", "flags": 0, "haystack": "http://www.google.com",
         "matches": True, "find": [True, 0, 21]},
        {"src": "DFACompilerTest.java:524-540", "pattern": "http://.+", "flags": 0, "haystack": "http://Γειά σου.com",
         "matches": True, "find": [True, 0, 19]},
        {"src": "DFACompilerTest.java:784-793", "pattern": "a*baa", "flags": 0, "haystack": "aaaabaa", "find_range": [3, 7],
         "find": [True, 3, 7]},
        {"src": "DFACompilerTest.java:801-813", "pattern": "(a*tgc*|t*acg*)*(cg)(a|t)*", "flags": 0, "haystack": "cgatgccgaa",
         "find_range": [6, 10], "find": [True, 6, 10]},
        {"src": "DFACompilerTest.java:827-842", "pattern": "a.*c", "flags": 0, "haystack": "abc\nc", "find": [True, 0, 3]},
        {"src": "DFACompilerTest.java:827-842", "pattern": "a.*c", "flags": 0x20, "haystack": "abc\nc", "find": [True, 0, 5]},
        {"src": "DFACompilerTest.java:827-842", "pattern": "a.*c", "flags": 0, "haystack": "abc", "matches": True},
    ]
    json.dump({"source": "needle-compiler/src/test/resources/matches.txt", "rows": rows, "inline": inline},
              open(OUT, "w"), ensure_ascii=True, indent=0)
    print(len(rows), "rows")


if __name__ == "__main__":
    main()
