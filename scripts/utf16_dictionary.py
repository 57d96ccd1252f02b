"""Dictionaries over UTF-16 rows (Java's native strings) next to the same rows as bytes: python scripts/utf16_dictionary.py [n_rows]
Prints find() / containedIn() ms and the algorithmic GB/s for c3 (1000 keywords of 3-5 chars), c3s (6-8 chars) and c3x (3000 keywords)."""
import sys, torch
sys.path.insert(0, ".")
from needle_amd.pattern import DFACompiler
from needle_amd import workload as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dev = "cuda"
for name, kw in (("c3", dict(n=1000)), ("c3s", dict(n=1000, min_len=6, max_len=8)), ("c3x", dict(n=3000, min_len=6, max_len=8))):
    words = W.keywords(kw.pop("n"), **kw)
    p = DFACompiler.compile("|".join(words), name)
    rows8 = torch.empty((n, 256), dtype=torch.uint8, device=dev)
    for s in range(0, n, 1 << 19):
        m = min(1 << 19, n - s)
        rows8[s:s + m] = W.keyword_batch(torch, words, s, m, 256, device=dev)
    rows16 = rows8.to(torch.int16)
    ref = None
    for label, rows in (("bytes ", rows8), ("utf-16", rows16)):
        for op, oname in ((p.find_batch, "find"), (p.contained_in_batch, "containedIn")):
            for _ in range(2): r = op(rows)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): r = op(rows)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            nb = rows.numel() * rows.element_size()
            extra = ""
            if oname == "find":
                got = (r[1].cpu(), r[2].cpu())
                if ref is None: ref = got
                else: extra = "  same as bytes: %s" % bool((got[0] == ref[0]).all() and (got[1] == ref[1]).all())
            print("%-4s %s %-12s %8.3f ms  %6.0f GB/s  (%.3f of 8 TB/s) mode %s%s" % (name, label, oname, ms, nb / ms / 1e6, nb / ms / 8e9,
                  p.info()["kernel_mode"]["forwards"], extra), flush=True)
    if name != "c3":
        # the same dictionary in Cyrillic letters (page 4 of the BMP): the byte program of the tables rebased to that page, everything else
        # in the text (the spaces between the words are page 0) narrowed to the page's substitute
        cw = ["".join(chr(0x0430 + ord(c) - 97) for c in w) for w in words]
        pc = DFACompiler.compile("|".join(cw), name + "-cyrillic")
        rc = torch.where((rows16 >= 97) & (rows16 <= 122), rows16 + (0x0430 - 97), rows16)
        for op, oname in ((pc.find_batch, "find"), (pc.contained_in_batch, "containedIn")):
            for _ in range(2): r = op(rc)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): r = op(rc)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            nb = rc.numel() * 2
            extra = ""
            if oname == "find":
                got = (r[1].cpu(), r[2].cpu())
                extra = "  same as the Latin dictionary on the Latin rows: %s" % bool((got[0] == ref[0]).all() and (got[1] == ref[1]).all())
            print("%-4s cyrillic utf-16 %-12s %8.3f ms  %6.0f GB/s  (%.3f of 8 TB/s)%s" % (name, oname, ms, nb / ms / 1e6, nb / ms / 8e9, extra), flush=True)
        del rc
    del rows8, rows16
