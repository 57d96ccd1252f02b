"""Random DICTIONARIES against the oracle: keyword unions big enough for the compressed automaton (mode 6) and its lengths program, with
keywords that are prefixes / suffixes / infixes of one another (states with a match pending that live on: END records, D_L rows as
default rows), planted at row ends and cut by ragged lengths.  python scripts/dictionary_fuzz.py <seed0> <n>   (FUZZ_MIN_LEN=5 | 7: dictionaries the n-gram filter takes; FUZZ_UTF16=1: the rows as UTF-16)
FUZZ_UTF16=3: keywords in three scripts (Latin, Cyrillic, CJK) over mixed-script UTF-16 rows -- the WIDE filter (with FUZZ_MIN_LEN >= 5).
(child processes: NEEDLE_MAX_PROG_LDS is read once per process)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, random, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from needle_amd.pattern import DFACompiler, unpack_bitmap
from test_compile_matches_txt import oracle_for
seed = int(sys.argv[1])
rng = random.Random(seed)
alpha = "abcdefghijklmnopqrstuvwxyz"[:rng.choice([8, 14, 26, 26])]
n_kw = rng.choice([150, 400, 900])
lo, hi = rng.choice([(3, 6), (5, 8), (6, 9), (2, 9)])
words = set()
while len(words) < n_kw:
    w = "".join(rng.choice(alpha) for _ in range(rng.randint(lo, hi)))
    words.add(w)
    r = rng.random()
    if r < 0.15 and len(w) > lo: words.add(w[:rng.randint(max(2, lo - 1), len(w) - 1)])      # a proper prefix
    elif r < 0.25 and len(w) > 3: words.add(w[rng.randint(1, len(w) - 2):])                    # a proper suffix
    elif r < 0.30: words.add(w + "".join(rng.choice(alpha) for _ in range(rng.randint(1, 3))))  # an extension
import os
import os
fuzz_min = int(os.environ.get("FUZZ_MIN_LEN", "0"))  # keep only keywords of at least this many chars (5: the n-gram filter's stride 2
if fuzz_min:                                            # becomes possible, 7: stride 4) and plant near misses (keyword tails) as well
    words = {w for w in words if len(w) >= fuzz_min}
    while len(words) < n_kw // 2:
        words.add("".join(rng.choice(alpha) for _ in range(rng.randint(fuzz_min, fuzz_min + 3))))
words = sorted(words)
rng.shuffle(words)
cyr = int(os.environ.get("FUZZ_UTF16", "0")) == 2  # 2: the dictionary in Cyrillic letters (page 4 of the BMP) over UTF-16 rows
mixed = int(os.environ.get("FUZZ_UTF16", "0")) == 3  # 3: keyword i in script i % 3 -- Latin, Cyrillic, CJK ideographs on 26 pages -- over mixed-script
SCRIPT = [lambda k: 97 + k, lambda k: 0x0430 + k, lambda k: 0x4E00 + 0x3FD * k]  # UTF-16 rows: no single page, the WIDE filter (FUZZ_MIN_LEN >= 5)
def enc(i):  # keyword i as the code units the text holds
    if mixed: return np.array([SCRIPT[i % 3](ord(c) - 97) for c in words[i]], dtype=np.uint16)
    return np.frombuffer(words[i].encode(), dtype=np.uint8)
rx = "|".join("".join(chr(0x0430 + ord(c) - 97) for c in w) for w in words) if cyr else "|".join(words)
if mixed: rx = "|".join("".join(chr(int(u)) for u in enc(i)) for i in range(len(words)))
p = DFACompiler.compile(rx, "t", 0)
o, _ = oracle_for(rx, 0)
pi = p.program_info("forwards", 1)
n, width = 20011, rng.choice([64, 112, 256])
nr = np.random.default_rng(seed)
noise = np.array([ord(c) for c in alpha + " "], dtype=np.uint8)
if mixed: noise = np.array([f(ord(c) - 97) for f in SCRIPT for c in alpha] + [32, 32, 32], dtype=np.uint16)
rows = nr.choice(noise, (n, width))
for r in range(0, n, 3):  # plant: anywhere, at the very end, cut by the end
    w = enc(int(nr.integers(len(words))))
    k = r % 9
    if k == 0: rows[r, width - len(w):] = w
    elif k == 3 and len(w) > 1: rows[r, width - len(w) + 1:] = w[:-1]
    else:
        at = int(nr.integers(0, width - len(w) + 1)); rows[r, at:at + len(w)] = w
if fuzz_min:  # near misses: a keyword's tail behind a wrong first char (passes the filter, matches nothing -- unless it does)
    for r in range(1, n, 3):
        wi = int(nr.integers(len(words)))
        w = enc(wi).copy()
        w[0] = SCRIPT[wi % 3](int(nr.integers(len(alpha)))) if mixed else ord(alpha[nr.integers(len(alpha))])
        at = int(nr.integers(0, width - len(w) + 1)); rows[r, at:at + len(w)] = w
utf16 = int(os.environ.get("FUZZ_UTF16", "0"))  # the same rows as UTF-16 (Java's strings), chars above 0xFF sprinkled over text and keywords:
if utf16:                                        # dictionaries with a filter take the byte program's filter kernel, the text narrowed on the fly
    rows = rows.astype(np.uint16)
    if cyr:
        low = (rows >= 97) & (rows <= 122)
        rows[low] += 0x0430 - 97
        rows[low & (nr.random(rows.shape) < 0.01)] += 0x0100   # page 5 under a keyword char's low byte
        rows[nr.random(rows.shape) < 0.01] = 0x0061             # page 0
    m = nr.random(rows.shape) < 0.01
    rows[m] = nr.integers(0x0100, 0xFFFF, size=int(m.sum()), dtype=np.uint16)
    rows[nr.random(rows.shape) < 0.01] |= 0x0100
    rows[nr.random(rows.shape) < 0.002] = 0x00FF
lens = nr.integers(0, width + 1, n).astype(np.uint32)
t = torch.from_numpy(rows.view(np.int16) if utf16 else rows).cuda()
tl = torch.from_numpy(lens.astype(np.int32)).cuda()
for l, dl in ((None, None), (lens, tl)):
    fw, fs, fe = p.find_batch(t, dl)
    of, ofs, ofe = o.batch_find(rows, l, threads=8)
    assert (unpack_bitmap(fw, n) == of).all(), ("find bitmap", seed)
    assert (fs.cpu().numpy() == ofs).all() and (fe.cpu().numpy() == ofe).all(), ("find start/end", seed)
    cur = torch.where(torch.from_numpy(of).cuda(), fe, torch.full_like(fe, -1))
    nw, ns, ne = p.find_next_batch(t, cur, dl)
    nw, ns, ne = unpack_bitmap(nw, n), ns.cpu().numpy(), ne.cpu().numpy()
    for i in range(0, n, 53):
        a = o.find_all(rows[i] if l is None else rows[i, :l[i]])
        if len(a) >= 2: assert nw[i] and (ns[i], ne[i]) == a[1], ("second find", seed, i)
        else: assert not nw[i], ("second find", seed, i)
    assert (unpack_bitmap(p.contained_in_batch(t, dl), n) == o.batch_contained_in(rows, l, threads=8)).all(), ("containedIn", seed)
    # every match of every row (dense slots, then the count pass) against the oracle's repeated find() on sampled rows
    want = {i: o.find_all(rows[i] if l is None else rows[i, :l[i]]) for i in range(0, n, 7)}
    most = max([len(w) for w in want.values()] + [1])
    for slots in (most + 2, 2):
        cnt, ast, aen, more = p.find_all_dense(t, slots, dl)
        cnt, ast, aen = cnt.cpu().numpy(), ast.cpu().numpy(), aen.cpu().numpy()
        for i, w in want.items():
            k = min(len(w), slots)
            assert cnt[i] == k and list(zip(ast[i, :k].tolist(), aen[i, :k].tolist())) == w[:k], ("find-all", seed, i, cnt[i], ast[i].tolist(), aen[i].tolist(), w[:6])
    cc = p.count_matches_batch(t, dl).cpu().numpy()
    for i, w in want.items():
        assert cc[i] == len(w), ("count pass", seed, i)
pf = p.prefilter_info("forwards", wide=mixed)
ft = p.find_all_transducer(1)  # (the budget is the process's: a transducer reported here is the one find-all walked in lock-step)
print("DICT-OK seed %d%s: %d keywords over %d letters, %d states, mode %d, lengths form %d, n-gram filter %s, find-all %s, %d of %d rows match" % (
    seed, " (UTF-16 rows, %d filter launches)" % p.prefilter_state("forwards")["filter_launches"] if utf16 else "", len(words), len(alpha), pi["n_states"], pi["mode"], pi["lengths_form"],
    ("stride %d run-up %d%s" % (pf["stride"], pf["warm"], " +level 2" if pf["on2"] else "")) if pf["on"] else "off",
    ("lock-step (%d states)" % ft["n_states"]) if ft is not None and not pf["on"] else "filter form" if pf["on"] else "one-pass kernel", int(of.sum()), n))
'''
if __name__ == "__main__":
    seed0, cnt = int(sys.argv[1]), int(sys.argv[2])
    bad = 0
    for seed in range(seed0, seed0 + cnt):
        env = dict(os.environ)
        if seed % 4 and not os.environ.get("FUZZ_DEFAULT_BUDGET"): env["NEEDLE_MAX_PROG_LDS"] = str([12000, 20000, 40000][seed % 3])  # smaller dictionaries into the compressed form too
        r = subprocess.run([sys.executable, "-c", CODE, str(seed)], env=env, capture_output=True, text=True, cwd=ROOT, timeout=900)
        line = [x for x in r.stdout.splitlines() if x.startswith("DICT-OK")]
        if line: print(line[0], "(LDS budget %s)" % env.get("NEEDLE_MAX_PROG_LDS", "default"))
        else:
            bad += 1
            print("FAILED seed", seed, r.stdout[-500:], r.stderr[-1500:])
    print("dictionary campaign done: %d seeds, %d failures" % (cnt, bad))
