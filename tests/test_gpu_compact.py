"""Compact result forms of find() (include/needle_hip.h: needle_find_compact_dev / _host, needle_find_packed16_host): the
matched rows only as {row, start, end} records in row order, and start / end as one dword per row -- against
needle_find_dev on the same rows and against the CPU oracle (Matcher.find() + start() + end(),
DFAClassBuilder.java:625-667)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_configs import compiled

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_records(recs, n_matched, words, m, s, e):
    from needle_amd.pattern import unpack_bitmap
    n = len(m)
    assert (unpack_bitmap(words, n) == m).all()
    rows = np.nonzero(m)[0]
    assert n_matched == len(rows) == len(recs)
    assert (recs["row"] == rows).all()                       # row order, matched rows only
    assert (recs["start"] == s[rows]).all() and (recs["end"] == e[rows]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("regex,width", [("[0-9]+", 256), ("[0-9]+", 48), ("Sherlock|Holmes|Watson|Irene|Adler|John|Baker", 128)])
@pytest.mark.parametrize("ragged", [False, True])
def test_compact_find_equals_find(regex, width, ragged):
    import torch
    from needle_amd import workload as W
    from needle_amd.pattern import Pattern
    p, o = compiled(regex)
    for n in (1, 63, 64 * 700 + 13, 300000):
        host = W.digits_batch(np, 17, n, width)
        if "Sherlock" in regex:
            host = host.copy()
            host[::5, 3:11] = np.frombuffer(b"Sherlock", dtype=np.uint8)
            host[2::7, width - 6:width] = np.frombuffer(b"Watson", dtype=np.uint8)
        lens = ((np.arange(n, dtype=np.uint64) * 2654435761) % (width + 1)).astype(np.uint32) if ragged else None
        m, s, e = o.batch_find(host, lens, threads=8)
        rows = torch.from_numpy(host).cuda()
        tl = None if lens is None else torch.from_numpy(lens.astype(np.int32)).cuda()
        words, recs, cnt = p.find_compact(rows, tl)
        torch.cuda.synchronize()
        k = int(cnt.item())
        r = recs[:k].cpu().numpy().view(np.uint32).reshape(-1, 2)
        rec = np.zeros(k, dtype=Pattern.MATCH_REC)
        rec["row"], rec["start"], rec["end"] = r[:, 0], r[:, 1] & 0xFFFF, r[:, 1] >> 16
        _check_records(rec, k, words, m, s, e)
        # a record buffer that is too small: the count is still the total, the first `cap` records are written
        if k > 3:
            small = (torch.empty((n + 63) // 64, dtype=torch.int64, device="cuda"), torch.full((3, 2), -1, dtype=torch.int32, device="cuda"),
                     torch.zeros(1, dtype=torch.int64, device="cuda"))
            p.find_compact(rows, tl, out=small)
            torch.cuda.synchronize()
            assert int(small[2].item()) == k and (small[1].cpu().numpy().view(np.uint32) == r[:3]).all()
        if n <= 64 * 700 + 13:  # the host forms (upload, scan, download of 1 bit per row + 8 bytes per matched row / 4 per row)
            hw, hrec = p.find_compact(host, lens)
            _check_records(hrec, len(hrec), hw, m, s, e)
            pw, se = p.find_packed16_host(host, lens)
            lo, hi = (se & 0xFFFF).astype(np.int64), (se >> 16).astype(np.int64)
            assert (np.where(lo == 0xFFFF, -1, lo) == s).all() and (np.where(hi == 0xFFFF, -1, hi) == e).all()
            assert (pw == hw).all()


CHUNKED = r'''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from needle_amd import workload as W
from needle_amd.pattern import unpack_bitmap
from test_gpu_configs import compiled
from test_gpu_compact import _check_records
p, o = compiled("[0-9]+")
n = 64 * 500 + 7
host = W.digits_batch(np, 3, n, 128)
m, s, e = o.batch_find(host, threads=8)
hw, hrec = p.find_compact(host)      # NEEDLE_HOST_CHUNK_BYTES = 1 MiB: 8192-row chunks, record row numbers run on across them
_check_records(hrec, len(hrec), hw, m, s, e)
pw, se = p.find_packed16_host(host)
assert (pw == hw).all() and ((se & 0xFFFF).astype(np.int64)[m] == s[m]).all() and ((se >> 16).astype(np.int64)[m] == e[m]).all() and (se[~m] == 0xFFFFFFFF).all()
print("CHUNKED-OK")
'''


@pytest.mark.gpu
def test_compact_host_forms_across_chunks():
    env = dict(os.environ, NEEDLE_HOST_CHUNK_BYTES=str(1 << 20))
    r = subprocess.run([sys.executable, "-c", CHUNKED], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert "CHUNKED-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_compact_find_rejects_rows_the_records_cannot_hold():
    import torch
    p, _ = compiled("[0-9]+")
    rows = torch.zeros((64, 65600), dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception):
        p.find_compact(rows)
