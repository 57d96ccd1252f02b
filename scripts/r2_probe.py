"""Round-2 probe (one gpurun call): where does C3's time go?  find() on the full rows vs only their first 128 / 64
bytes (what a survivor-deferring kernel would walk in place), liveness of the rows, and the sparse-match variants."""
import ctypes, sys, time, torch
sys.path.insert(0, ".")
import bench
from needle_amd import workload as W, _lib
from needle_amd._lib import BatchView
from needle_amd.pattern import DFACompiler, unpack_bitmap

n = 10_000_000
L = _lib.lib()

def timed(fn, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def run_view(p, op, rows, row_len):
    nr, stride = rows.shape
    v = BatchView()
    v.rows, v.char_width, v.n_rows, v.row_stride, v.row_len = rows.data_ptr(), rows.element_size(), nr, stride, row_len
    words = torch.empty((nr + 63) // 64, dtype=torch.int64, device="cuda")
    st = torch.empty(nr, dtype=torch.int32, device="cuda"); en = torch.empty(nr, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    if op == "find":
        f = lambda: L.needle_find_dev(p._h, ctypes.byref(v), words.data_ptr(), st.data_ptr(), en.data_ptr(), s)
    else:
        f = lambda: L.needle_contained_in_dev(p._h, ctypes.byref(v), words.data_ptr(), s)
    return timed(f), words, st, en

words = W.keywords(1000)
p = DFACompiler.compile("|".join(words), "k")
rows = bench.make_rows("c3", words, 0, n, "cuda")
print("C3 dense: states", p.info()["n_states"]["forwards"], "mode", p.info()["kernel_mode"]["forwards"])
for op in ("find", "contained_in"):
    for rl in (256, 192, 128, 64):
        ms, w, st, en = run_view(p, op, rows, rl)
        print("  %-12s row_len %3d: %.3f ms" % (op, rl, ms))
ms, w, st, en = run_view(p, "find", rows, 256)
m = torch.from_numpy(unpack_bitmap(w, n)).cuda()
print("  matched %.4f" % m.float().mean().item())
for x in (16, 32, 64, 96, 128, 192, 256):
    # a row is resolved once the walk has died: approximately end + 2 <= x for matched rows
    print("  rows with end+2 <= %3d: %.4f" % (x, (m & (en + 2 <= x)).float().mean().item()))
del rows
for nk, lo, hi in ((1000, 6, 8), (300, 6, 8)):
    t0 = time.time()
    ws = W.keywords(nk, min_len=lo, max_len=hi)
    ps = DFACompiler.compile("|".join(ws), "ks")
    rows = bench.make_rows("c3", ws, 0, n, "cuda")
    inf = ps.info()
    print("C3 sparse %d x %d..%d: states %d mode %d (compile+gen %.1fs)" % (nk, lo, hi, inf["n_states"]["forwards"], inf["kernel_mode"]["forwards"], time.time() - t0))
    for op in ("find", "contained_in"):
        ms, w, st, en = run_view(ps, op, rows, 256)
        print("  %-12s %.3f ms  matched %.4f" % (op, ms, unpack_bitmap(w, n).mean()))
    del rows
