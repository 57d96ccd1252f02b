"""find() two ways on the LDS-table automata: the "lengths" form (the forward automaton refined until its stop state
remembers the match's length -- start = end - length, no indexBackwards; needle_amd/csrc/needle_lower.h, the per-state
generalisation of DFAClassBuilder.java:640-656) and the reference's own forward + backward walks (DFAClassBuilder.java:
616-659; NEEDLE_FIND_LENGTHS=0).  Both must give the reference's (found, start, end) on full, ragged, short and UTF-16 rows
and from per-row cursors.  The switch is read once per process, so each form runs in a child."""
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
want_form = int(sys.argv[1])
big = W.keywords(1000, min_len=6, max_len=8)  # 4439 states: the compressed whole-automaton form (mode 6) and ITS lengths program
cases = [("|".join(W.keywords(300)), 2, None), ("|".join(W.keywords(60)) + "|ab|abc|bc|bcd", 1, None), ("(foo|fo|o)(bar|ba|r)x?q", None, None),
         ("[a-c]{2,4}d|xy", None, None), ("|".join(big), 6, big)]
for rx, mode, plant in cases:
    p = DFACompiler.compile(rx, "t", 0)
    o, _ = oracle_for(rx, 0)
    for cw in (1, 2):
        pi = p.program_info("forwards", cw)
        if pi["mode"] in (1, 2, 4, 6):
            assert pi["lengths_form"] == want_form, (rx[:40], pi)
    if mode is not None:
        assert p.program_info("forwards", 1)["mode"] == mode, p.program_info("forwards", 1)
    words = plant or W.keywords(300)
    n = 30011
    rows = W.keyword_batch(torch, words, 7, n, 256, device="cuda")
    if plant:  # a keyword ending exactly at the row's end, and one cut by it: the END record / the frozen state of ragged rows
        kw = torch.tensor([ord(c) for c in plant[5]], dtype=torch.uint8, device="cuda")
        rows[::97, 256 - len(kw):] = kw
        rows[50::97, 256 - len(kw) + 1:] = kw[:-1]
    if mode is None:  # a small alphabet, so that the short patterns match often and at every offset
        g = torch.Generator(device="cuda"); g.manual_seed(11)
        lut = torch.tensor([ord(c) for c in "abcdfoqrxy"], dtype=torch.uint8, device="cuda")
        rows = lut[torch.randint(0, 10, (n, 256), device="cuda", generator=g)]
    host = rows.cpu().numpy()
    lens = (torch.arange(n, device="cuda", dtype=torch.int64) * 2654435761 % 257).to(torch.int32)
    hl = lens.cpu().numpy().astype(np.uint32)
    for r, l, h, hlen in ((rows, None, host, None), (rows, lens, host, hl), (rows.to(torch.int16), lens, host.astype(np.uint16), hl),
                          (rows[:, :32].contiguous(), None, np.ascontiguousarray(host[:, :32]), None),
                          (rows[:, :48].contiguous(), (lens % 49).to(torch.int32), np.ascontiguousarray(host[:, :48]), hl % 49)):
        fw, fs, fe = p.find_batch(r, l)
        of, ofs, ofe = o.batch_find(h, hlen, threads=4)
        assert (unpack_bitmap(fw, n) == of).all(), rx[:40]
        assert (fs.cpu().numpy() == ofs).all() and (fe.cpu().numpy() == ofe).all(), rx[:40]
        assert of.sum() > n // 50 or (plant and r.shape[1] < 64)  # (6..8-char keywords are rare in 32- and 48-char windows)
        # the next find() of every row, from the cursor the first one left (DFAClassBuilder.java:616-625)
        cur = torch.where(torch.from_numpy(of).cuda(), fe, torch.full_like(fe, -1))
        nw, ns, ne = p.find_next_batch(r, cur, l)
        nw, ns, ne = unpack_bitmap(nw, n), ns.cpu().numpy(), ne.cpu().numpy()
        for i in range(0, n, 37):
            row = h[i] if hlen is None else h[i, :hlen[i]]
            all_ = o.find_all(row)
            if len(all_) >= 2:
                assert nw[i] and (ns[i], ne[i]) == all_[1], (rx[:40], i, all_[:3], ns[i], ne[i])
            else:
                assert not nw[i], (rx[:40], i)
print("FIND-FORMS-OK")
'''


@pytest.mark.gpu
@pytest.mark.parametrize("form", [1, 0], ids=["lengths", "backward-walk"])
def test_find_forms_match_oracle(form):
    env = dict(os.environ, NEEDLE_FIND_LENGTHS=str(form))
    r = subprocess.run([sys.executable, "-c", CODE, str(form)], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert "FIND-FORMS-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
