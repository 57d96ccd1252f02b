"""A dictionary at the reference's own limit (DFACompiler.java:76-83: 16 383 states): 3000 keywords of 6..8 chars = 12 270 states,
690 KB as a table -- an automaton that fits the LDS in NO form.  containedIn() / find() / find-all run behind the n-gram candidate
filter with the candidates' walks out of HBM / L2 (needle_lower.cpp lower_filter_hbm, ngram_kernel<.., MODE_GLOBAL, ..>); with
NEEDLE_PREFILTER=0 they take hot rows + HBM table in the scan kernels.  Both against the CPU oracle (oracle/needle_walk.c:
DFAClassBuilder.java:335-471, 625-659, 956-1025), bit for bit: random text, keywords at both ends of a row and cut by it, ragged
rows, strides that are not whole batches of units, and near-miss text (rows built from keyword prefixes -- the text on which the
hot-rows walk collapses)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler, unpack_bitmap
from test_compile_matches_txt import oracle_for
level = int(sys.argv[1])
words = W.keywords(3000, min_len=6, max_len=8)
rx = "|".join(words)
p = DFACompiler.compile(rx, "t", 0)
o, _ = oracle_for(rx, 0)
inf = p.info()
assert inf["n_states"]["forwards"] > 12000 and inf["kernel_mode"]["forwards"] == 5, inf   # hot rows + HBM table: no LDS form fits
fi, ci = p.prefilter_info("forwards"), p.prefilter_info("contained_in")
if level == 0:
    assert not fi["on"] and not ci["on"]
else:
    assert fi["on"] and ci["on"] and fi["mode"] == 3 and fi["stride"] == 2, (fi, ci)
dev = "cuda"
def check(rows, lens, tag):
    n = rows.shape[0]
    host = rows.cpu().numpy()
    hl = None if lens is None else lens.cpu().numpy().astype(np.uint32)
    fw, fs, fe = p.find_batch(rows, lens)
    cw = p.contained_in_batch(rows, lens)
    pw, pk = p.find_packed16_batch(rows, lens)
    torch.cuda.synchronize()
    of, ofs, ofe = o.batch_find(host, hl, threads=8)
    oc = o.batch_contained_in(host, hl, threads=8)
    assert (unpack_bitmap(fw, n) == of).all(), (tag, "find bitmap")
    bs = np.nonzero((fs.cpu().numpy() != ofs) | (fe.cpu().numpy() != ofe))[0]
    assert bs.size == 0, (tag, "start/end", bs[:5], fs.cpu().numpy()[bs[:5]], ofs[bs[:5]], fe.cpu().numpy()[bs[:5]], ofe[bs[:5]])
    assert (unpack_bitmap(cw, n) == oc).all(), (tag, "containedIn")
    pkv = pk.cpu().numpy().view(np.uint32)
    assert (unpack_bitmap(pw, n) == of).all() and ((pkv & 0xFFFF).astype(np.int32)[of] == ofs[of]).all() and ((pkv >> 16).astype(np.int32)[of] == ofe[of]).all(), (tag, "packed16")
    want = {i: o.find_all(host[i] if hl is None else host[i, :hl[i]]) for i in range(0, n, 5)}
    most = max([len(w) for w in want.values()] + [1])
    slots = most + 1
    counts, st, en, more = p.find_all_dense(rows, slots, lens)
    torch.cuda.synchronize()
    counts, st, en = counts.cpu().numpy(), st.cpu().numpy(), en.cpu().numpy()
    for i, w in want.items():
        k = min(len(w), slots)
        assert counts[i] == k and list(zip(st[i, :k].tolist(), en[i, :k].tolist())) == w[:k], (tag, "find-all", i, counts[i], st[i], en[i], w[:6])
    cnt = p.count_matches_batch(rows, lens).cpu().numpy()
    assert (np.minimum(cnt, slots) == counts).all(), (tag, "count pass")
    return int(of.sum())
kw = [torch.tensor([ord(c) for c in w], dtype=torch.uint8, device=dev) for w in words[:8]]
total = 0
for stride, n in ((256, 64 * 300 + 13), (64, 64 * 300 + 7), (192, 64 * 100 + 63), (1024, 64 * 20 + 1), (272, 64 * 60 + 33)):
    rows = W.keyword_batch(torch, words, 3, n, stride, device=dev)
    k0, k1, k2 = kw[0], kw[1], kw[2]
    rows[::11, stride - len(k0):] = k0                  # a keyword that ends with the row
    rows[5::11, stride - len(k1) + 1:] = k1[:-1]        # one that the row's end cuts
    rows[7::11, :len(k2)] = k2                          # one at the very start
    rows[9::11, 1:1 + len(k0)] = k0
    total += check(rows, None, (stride, n, "full"))
    lens = (torch.arange(n, device=dev, dtype=torch.int64) * 2654435761 % (stride + 1)).to(torch.int32)
    total += check(rows, lens, (stride, n, "ragged"))
# near misses: rows made of keywords with a char changed here and there -- lanes deep in the automaton, candidates everywhere
g = torch.Generator(device=dev); g.manual_seed(5)
n = 64 * 100 + 5
wt = torch.zeros((len(words), 16), dtype=torch.uint8, device=dev) + 32
for i, w in enumerate(words):
    wt[i, :len(w)] = torch.tensor([ord(c) for c in w], dtype=torch.uint8, device=dev)
pick = torch.randint(0, len(words), (n, 16), device=dev, generator=g)
rows = wt[pick].reshape(n, 256).clone()
flip = torch.rand((n, 256), device=dev, generator=g) < 0.08
rows = torch.where(flip, torch.full_like(rows, ord("q")), rows)
total += check(rows, None, ("near-miss",))
assert total > 5000, total
print("BIG-DICTIONARY-OK", level, total)
'''


@pytest.mark.gpu
@pytest.mark.parametrize("level", [1, 0], ids=["filter + HBM-table walks", "off: hot rows + HBM table"])
def test_dictionary_at_the_state_limit_vs_oracle(level):
    r = subprocess.run([sys.executable, "-c", CODE, str(level)], env=dict(os.environ, NEEDLE_PREFILTER=str(level)), capture_output=True, text=True,
                       timeout=1800, cwd=ROOT)
    assert "BIG-DICTIONARY-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
