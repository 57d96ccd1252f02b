#!/usr/bin/env python3
"""Generates the Unicode data the regex front end needs for UNICODE_CHARACTER_CLASS (\\d \\s \\w) and UNICODE_CASE:

    needle_amd/csrc/needle_unicode_tables.h   (C++ data, included by needle_regex.cpp)
    oracle/unicode_bmp.json                   (the same data for the Python restatement oracle/needle_compile.py)

The reference builds these sets at class-load time from the JDK's java.lang.Character database
(RegexParser.java:40-63: isDigit, isWhitespace, isAlphabetic / getType; :277-291: toUpperCase / toLowerCase), i.e. from
whatever Unicode version the running JDK carries.  There is no JDK in the build container; this script reads the
Unicode Character Database that ships with perl (Unicode::UCD, the version is recorded in both outputs; 13.0.0 = what
JDK 15..18 carry) and restates the java.lang.Character predicates on it:

    isDigit(c)       general category Nd
    isWhitespace(c)  Zs | Zl | Zp except the no-break spaces U+00A0 U+2007 U+202F, plus U+0009..U+000D, U+001C..U+001F
    isAlphabetic(c)  the derived property Alphabetic (Lu Ll Lt Lm Lo Nl + Other_Alphabetic)
    word (\\w)        Alphabetic | Mn | Me | Mc | Nd | Pc          (RegexParser.java:51-57; no JOIN_CONTROL there)
    toUpperCase / toLowerCase(char)   the SIMPLE case mappings of UnicodeData.txt (no SpecialCasing)

Only the BMP matters (Java chars), and U+FFFF is excluded as in the reference's loops (`candidate < Character.MAX_VALUE`).
Run in the build container:  python scripts/gen_unicode_tables.py
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PERL = r'''
use strict; use warnings;
use Unicode::UCD qw(prop_invmap prop_invlist);
print "version ", Unicode::UCD::UnicodeVersion(), "\n";
my @l = prop_invlist("Alphabetic");
print "alpha @l\n";
my ($list, $map, $format, $default) = prop_invmap("General_Category");
for my $i (0..$#$list) { print "gc $list->[$i] $map->[$i]\n"; }
for my $p (qw(Simple_Uppercase_Mapping Simple_Lowercase_Mapping)) {
  my ($list, $map, $format, $default) = prop_invmap($p);
  die "unexpected format $format" unless $format eq "a";
  for my $i (0..$#$list) { my $m = $map->[$i]; die "list mapping" if ref $m; print "map $p $list->[$i] $m\n"; }
}
'''


def inversion_to_flags(points, n=0x10000):
    flags = [False] * n
    for i in range(0, len(points), 2):
        lo = points[i]
        hi = points[i + 1] if i + 1 < len(points) else 0x110000
        for c in range(lo, min(hi, n)):
            flags[c] = True
    return flags


def ranges_of(flags):
    out, c, n = [], 0, len(flags)
    while c < n:
        if flags[c]:
            s = c
            while c + 1 < n and flags[c + 1]:
                c += 1
            out.append((s, c))
        c += 1
    return out


def main():
    txt = subprocess.run(["perl", "-e", PERL], check=True, capture_output=True, text=True).stdout.splitlines()
    version = None
    gc_pts, maps, alpha = [], {"Simple_Uppercase_Mapping": [], "Simple_Lowercase_Mapping": []}, None
    for line in txt:
        f = line.split()
        if f[0] == "version":
            version = f[1]
        elif f[0] == "alpha":
            alpha = inversion_to_flags([int(x) for x in f[1:]])
        elif f[0] == "gc":
            gc_pts.append((int(f[1]), f[2]))
        elif f[0] == "map":
            maps[f[1]].append((int(f[2]), int(f[3])))
    gc = ["Cn"] * 0x10000
    for i, (start, cat) in enumerate(gc_pts):
        end = gc_pts[i + 1][0] if i + 1 < len(gc_pts) else 0x110000
        for c in range(start, min(end, 0x10000)):
            gc[c] = cat

    def simple(map_pts):
        # prop_invmap format "a": within a range starting at list[i] with map[i] != 0 the mapping is map[i] + (c - list[i]);
        # 0 = the code point maps to itself
        m = list(range(0x10000))
        for i, (start, base) in enumerate(map_pts):
            end = map_pts[i + 1][0] if i + 1 < len(map_pts) else 0x110000
            if base == 0:
                continue
            for c in range(start, min(end, 0x10000)):
                m[c] = base + (c - start)
        return m

    upper = simple(maps["Simple_Uppercase_Mapping"])
    lower = simple(maps["Simple_Lowercase_Mapping"])
    N = 0xFFFF  # `candidate < Character.MAX_VALUE`: U+FFFF is never a member
    nobreak = {0x00A0, 0x2007, 0x202F}
    ctrl_ws = set(range(0x09, 0x0E)) | set(range(0x1C, 0x20))
    digit = [c < N and gc[c] == "Nd" for c in range(0x10000)]
    space = [c < N and ((gc[c] in ("Zs", "Zl", "Zp") and c not in nobreak) or c in ctrl_ws) for c in range(0x10000)]
    word = [c < N and (alpha[c] or gc[c] in ("Mn", "Me", "Mc", "Nd", "Pc")) for c in range(0x10000)]
    # a BMP char whose simple mapping leaves the BMP cannot be represented by Character.toUpperCase(char)'s (char)
    # cast; Unicode has no such simple mapping, checked here
    assert all(0 <= upper[c] < 0x10000 and 0 <= lower[c] < 0x10000 for c in range(0x10000))
    up_pairs = [(c, upper[c]) for c in range(0x10000) if upper[c] != c]
    lo_pairs = [(c, lower[c]) for c in range(0x10000) if lower[c] != c]
    data = {"unicode_version": version, "digit": ranges_of(digit), "space": ranges_of(space), "word": ranges_of(word),
            "upper": up_pairs, "lower": lo_pairs}
    # spot checks against facts stated in the Unicode standard / java.lang.Character's documentation
    assert digit[0x0660] and digit[0xFF10] and not digit[0x00B2] and not digit[0x2160]
    assert space[0x2028] and space[0x1680] and not space[0x00A0] and space[0x1F] and not space[0x85]
    assert word[ord("_")] and word[0x0300] and word[0x2160] and word[0x24B6] and not word[ord("-")]
    assert upper[0x00B5] == 0x039C and lower[0x0130] == 0x0069 and upper[0x0131] == 0x0049 and upper[0x00DF] == 0x00DF
    assert lower[0x212A] == 0x006B and upper[0x017F] == 0x0053 and lower[0x1E9E] == 0x00DF and upper[0x1F80] == 0x1F88

    with open(os.path.join(ROOT, "oracle", "unicode_bmp.json"), "w") as f:
        json.dump(data, f, separators=(",", ":"))
        f.write("\n")

    def c_ranges(name, rs):
        body = ",".join("{0x%X,0x%X}" % r for r in rs)
        lines, cur = [], ""
        for tok in body.split("},"):
            tok = tok if tok.endswith("}") else tok + "}"
            if len(cur) + len(tok) + 1 > 116:
                lines.append(cur)
                cur = ""
            cur += tok + ","
        lines.append(cur.rstrip(","))
        return "static const uint16_t %s[][2] = {\n    %s\n};\nstatic const int %s_n = %d;\n" % (name, "\n    ".join(lines), name, len(rs))

    with open(os.path.join(ROOT, "needle_amd", "csrc", "needle_unicode_tables.h"), "w") as f:
        f.write("// GENERATED by scripts/gen_unicode_tables.py from the Unicode Character Database %s (perl Unicode::UCD) -- do not edit.\n" % version)
        f.write("// What java.lang.Character answers for the BMP under that Unicode version (what JDK 15..18 carry for 13.0.0): the sets\n")
        f.write("// behind \\d \\s \\w under UNICODE_CHARACTER_CLASS (RegexParser.java:40-63) and the simple case mappings behind\n")
        f.write("// UNICODE_CASE (RegexParser.java:277-291).  Inclusive [first, last] ranges; U+FFFF is never a member.\n")
        f.write("#pragma once\n#include <stdint.h>\nnamespace needle_unicode {\n")
        f.write('static const char kUnicodeVersion[] = "%s";\n' % version)
        f.write(c_ranges("kDigit", data["digit"]))
        f.write(c_ranges("kSpace", data["space"]))
        f.write(c_ranges("kWord", data["word"]))
        f.write("// (code unit, Character.toUpperCase(code unit)) where they differ\n")
        f.write(c_ranges("kUpper", up_pairs))
        f.write("// (code unit, Character.toLowerCase(code unit)) where they differ\n")
        f.write(c_ranges("kLower", lo_pairs))
        f.write("} // namespace needle_unicode\n")
    print("Unicode %s: digit %d ranges, space %d, word %d; upper %d, lower %d pairs" % (
        version, len(data["digit"]), len(data["space"]), len(data["word"]), len(up_pairs), len(lo_pairs)), file=sys.stderr)


if __name__ == "__main__":
    main()
