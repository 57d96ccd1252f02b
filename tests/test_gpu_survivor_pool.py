"""The survivor pool of the tiled scan kernel (needle_kernels.hip): groups whose unresolved rows are few hand them to
their wave's pool; pool steps walk further 128-byte lines of up to 64 pooled rows gathered by per-lane addresses.
The pool only exists in the big-table kernels (64-byte tiles), so these tests use the 1000-keyword union (a 87 KB
uint16 table) and its sparse-match variant (the compressed automaton in LDS -- dense rows + exception records -- and,
with NEEDLE_SPARSE=0, hot rows in LDS + HBM table): find / containedIn / matches, full and
ragged rows, rows of several lines, against the CPU oracle; and every deferral threshold in a child process."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from needle_amd import workload as W
from needle_amd.pattern import DFACompiler, unpack_bitmap
from test_compile_matches_txt import oracle_for
lo, hi, want_mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
words = W.keywords(1000, min_len=lo, max_len=hi)
rx = "|".join(words)
p = DFACompiler.compile(rx, "t", 0)
assert p.info()["kernel_mode"]["forwards"] == want_mode, p.info()
o, _ = oracle_for(rx, 0)
for n, width in ((60000, 256), (9000, 1024), (5000, 192)):
    rows = W.keyword_batch(torch, words, 11, n, width, device="cuda")
    host = rows.cpu().numpy()
    lens = ((np.arange(n, dtype=np.uint64) * 2654435761) % (width + 1)).astype(np.uint32)
    lens[::7] = width
    for l in (None, lens):
        tl = None if l is None else torch.from_numpy(l.astype(np.int32)).cuda()
        fw, fs, fe = p.find_batch(rows, tl)
        of, ofs, ofe = o.batch_find(host, l, threads=8)
        assert (unpack_bitmap(fw, n) == of).all(), ("find bitmap", n, width, l is None)
        assert (fs.cpu().numpy() == ofs).all() and (fe.cpu().numpy() == ofe).all(), ("find start/end", n, width, l is None)
        assert (unpack_bitmap(p.contained_in_batch(rows, tl), n) == o.batch_contained_in(host, l, threads=8)).all(), ("containedIn", n, width)
        assert (unpack_bitmap(p.matches_batch(rows, tl), n) == o.batch_matches(host, l, threads=8)).all(), ("matches", n, width)
print("POOL-OK")
'''


@pytest.mark.gpu
@pytest.mark.parametrize("defer", ["0", "1", "16", "32"])
@pytest.mark.parametrize("lo,hi,mode", [(3, 5, 2), (6, 8, 6), (6, 8, 5)])
def test_survivor_pool_matches_oracle(defer, lo, hi, mode):
    env = dict(os.environ, NEEDLE_DEFER=defer)
    if mode == 5:
        env["NEEDLE_SPARSE"] = "0"  # the hot-rows fallback of automata the compressed form cannot hold
    r = subprocess.run([sys.executable, "-c", CODE, str(lo), str(hi), str(mode)], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert "POOL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("lo,hi,mode,dict_mode", [(6, 8, 6, "1"), (3, 5, 2, "2")])
def test_two_sets_per_wave_kernel_matches_oracle(lo, hi, mode, dict_mode):
    """needle_dict.hip (measurement builds only -- scripts/build_tuning.sh -- and there opt-in, NEEDLE_DICT): two 64-row sets per
    wave, 32-byte tiles, find()'s starts by a separate backward pass -- the 60 000 x 256 full-row batch of the child takes it
    (the ragged / short / odd-stride batches the ordinary kernel).  Skipped where the measurement library is not built."""
    tuning = os.path.join(ROOT, "needle_amd", "libneedle_hip_tuning.so")
    if not os.path.exists(tuning):
        pytest.skip("libneedle_hip_tuning.so not built (scripts/build_tuning.sh): the two-sets kernel is not part of the product")
    env = dict(os.environ, NEEDLE_DICT=dict_mode, NEEDLE_LIB=tuning)
    r = subprocess.run([sys.executable, "-c", CODE, str(lo), str(hi), str(mode)], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert "POOL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
