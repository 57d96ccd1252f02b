#!/usr/bin/env python3
"""bench.py -- the BASELINE.json metric on MI355X: GB/s of haystack scanned (+ rows/s, matches/s) by the DFA
table-walk hot path on the 10M x 256-char synthetic batch.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c3s|c5|c5w] [--also c3,c3s,c5,c5w|none]
                    [--rows R] [--scaling strong|weak] [--graph off|scan]

One "step" = one pass of the hot path over the whole device-resident batch: one kernel launch per GPU and, for
N > 1, the gathers that bring the results to rank 0 (bitmap: one RCCL all-gather; find(): plus a fan-in of start /
end to rank 0).  N > 1 is launched by the driver with torch.distributed.run, one rank per GPU; the SAME 10M-row batch
is sharded by contiguous row blocks (config C4: strong scaling; `--scaling weak` gives every GPU its own 10M rows
instead); there is no data-path collective.  Rank 0 prints ONE JSON line.

  headline    `--workload` (default c2: '[0-9]+' containedIn(), the configuration BASELINE.json's metric is quoted
              on).  value = whole-job algorithmic GB/s (input bytes + result bytes, per step, all
              GPUs) over the barrier-bracketed wall time of exactly K steps (max over ranks), inputs resident in HBM.
  workloads   the other BASELINE configs measured the same way in the same run (`--also`): c3 (union of 1k keywords,
              find), c3s (its sparse-match variant: keywords of 6..8 chars, only the planted 25 % of the rows match,
              every lane stays live to the end of its row), c5 (BMP class regex over UTF-16, find), c5w (its wide variant: per-script
              runs in sequence -- 30 char classes, 33 states: UTF-16 rows through the two-level page map into an LDS table).
  N > 1       the gathers are issued by the library itself (needle_multi_*: its own RCCL communicator, ONE call per
              step, queued on the scan's stream right behind the kernel); torch.distributed only carries the
              communicator id, the barrier and the max-over-ranks of the clock.  Reported beside the step time: scan_ms,
              gather_ms (blocking, separately).
  roofline    the scan kernel alone: algorithmic bytes per launch / mean launch duration from HIP events recorded on
              the launch stream inside the timed region; peak = 8 TB/s HBM3E.
  cpu_baseline  the CPU oracle (oracle/needle_walk.c, a port of the reference's generated loops -- NOT the JVM
              bytecode path: no JDK on the box) on a bounded sample of the same rows: all usable host cores, and one.
"""
import argparse
import hashlib
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL across processes)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
ENGINE_CLOCK_MHZ = 2400.0  # MI355X peak engine clock (same guide); chars/clk/CU is quoted against it
KERNEL_SOURCES = ["needle_scan.h", "needle_kernels.hip", "needle_stripe.hip", "needle_walk.h", "needle_device.h", "needle_lower.cpp",
                  "needle_ngram.h", "needle_ngram.hip", "needle_ngram_host.cpp", "needle_find_all.hip", "needle_find_all_ls.hip"]


def kernel_source_sha():
    """Identity of the kernels a profile was taken with: sha256 over the device sources + the lowering."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "needle_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_PATTERNS = {}


def make_pattern(workload):
    """(pattern, label, planted words) of a workload; compiled once per process and regex (c3s16 / c3x16 scan the c3s / c3x dictionaries:
    a 3000-keyword compile takes ~30 s)."""
    key = {"c3s16": "c3s", "c3x16": "c3x"}.get(workload, workload)
    if key not in _PATTERNS:
        _PATTERNS[key] = _make_pattern(key)
    p, _, words = _PATTERNS[key]
    return p, _make_label(workload, _PATTERNS[key][1]), words


def _make_label(workload, label):
    if workload == "c3s16":
        return "union-of-1k-keywords (6..8 chars) find() over UTF-16 rows (Java's strings): the byte program's n-gram filter, text narrowed as it is loaded"
    if workload == "c3x16":
        return ("union-of-3k-keywords (12 270 states) find() over UTF-16 rows (Java's strings): the byte program's n-gram filter, text narrowed as it is "
                "loaded, candidates' walks out of L2")
    return label


def _make_pattern(workload):
    from needle_amd import workload as W
    from needle_amd.pattern import DFACompiler
    if workload == "c2":
        return DFACompiler.compile("[0-9]+", "DigitPlus"), "'[0-9]+' containedIn()", None
    if workload == "c3":
        words = W.keywords(1000)
        return DFACompiler.compile("|".join(words), "Keywords1k"), "union-of-1k-keywords (3..5 chars) find()", words
    if workload == "c3s":
        words = W.keywords(1000, min_len=6, max_len=8)
        return (DFACompiler.compile("|".join(words), "Keywords1kSparse"),
                "union-of-1k-keywords, sparse-match variant (6..8 chars: only the planted 25 % of the rows match) find()", words)
    if workload == "c3s16":
        words = W.keywords(1000, min_len=6, max_len=8)
        return (DFACompiler.compile("|".join(words), "Keywords1kSparse"),
                "union-of-1k-keywords (6..8 chars) find() over UTF-16 rows (Java's strings): the byte program's n-gram filter, text narrowed as it is loaded", words)
    if workload == "c3x16":
        words = W.keywords(3000, min_len=6, max_len=8)
        return (DFACompiler.compile("|".join(words), "Keywords3k"),
                "union-of-3k-keywords (12 270 states) find() over UTF-16 rows (Java's strings): the byte program's n-gram filter, text narrowed as it is "
                "loaded, candidates' walks out of L2", words)
    if workload == "c3u":
        kws = W.keywords(1000, min_len=6, max_len=8)
        words = [w + str(10 + 7 * i % 90) for i, w in enumerate(kws)]  # what the rows hold: a keyword and two digits
        return (DFACompiler.compile("(" + "|".join(kws) + ")[0-9]+", "Keywords1kDigits"),
                "(union-of-1k-keywords)[0-9]+ -- no bounded match length: find() behind the n-gram filter, starts by backward walks", words)
    if workload == "c3m16":
        words = W.keywords_mixed(1000)
        return (DFACompiler.compile("|".join(words), "KeywordsMixed3k"),
                "union of 1000 Latin + 1000 Cyrillic + 1000 CJK keywords (6..8 code units; ~25 pages of the BMP) find() over mixed-script UTF-16 rows: "
                "the WIDE n-gram filter (windows of four 16-bit code units), candidates' walks on the UTF-16 table out of L2", words)
    if workload == "c3x":
        words = W.keywords(3000, min_len=6, max_len=8)
        return (DFACompiler.compile("|".join(words), "Keywords3k"),
                "union-of-3k-keywords at the reference's state limit (6..8 chars: 12 270 states <= 16 383, DFACompiler.java:76-83; no LDS form fits) find()", words)
    if workload == "c5":
        return DFACompiler.compile(W.script_regex(), "ScriptRuns"), "BMP char-class regex find() over UTF-16", None
    if workload == "c5w":
        return (DFACompiler.compile(W.scriptseq_regex(), "ScriptSeq"),
                "BMP multi-class regex (per-script runs in sequence: 30 classes, 33 states) find() over UTF-16", None)
    raise SystemExit("unknown workload " + workload)


_ARENA = {}


def batch_buffer(n_rows, dtype, device):
    """The resident batch lives in ONE device buffer for the whole process, sized for the largest workload (10M x 256 UTF-16 rows: 5.12 GB)
    and allocated before anything else; every workload's rows are a view of it.  A production host keeps its shard buffers resident the same
    way -- and the measurement stops depending on where the allocator happens to put a fresh 5 GB block after four other batches have come
    and gone: C5's kernel ran at 0.870-0.881 ms in a fresh process and anywhere between 0.877 and 0.943 ms as the fifth workload of one
    (same box, same lease, rocprofv3 says the same)."""
    import torch
    nbytes = n_rows * 256 * (2 if dtype == torch.int16 else 1)
    key = str(device)
    if key not in _ARENA or _ARENA[key].numel() < nbytes:
        _ARENA.pop(key, None)
        torch.cuda.empty_cache()
        _ARENA[key] = torch.empty(max(nbytes, n_rows * 512), dtype=torch.uint8, device=device)
    return _ARENA[key][:nbytes].view(dtype).view(n_rows, 256)


def make_rows(workload, words, row0, n_rows, device):
    """Shard [row0, row0 + n_rows) of the synthetic batch, generated on the GPU in slabs (into the process's one batch buffer)."""
    import torch
    from needle_amd import workload as W
    dtype = torch.int16 if workload in ("c5", "c5w", "c3s16", "c3x16", "c3m16") else torch.uint8
    out = batch_buffer(n_rows, dtype, device)
    slab = 1 << 19
    for s in range(0, n_rows, slab):
        n = min(slab, n_rows - s)
        if workload == "c2":
            out[s:s + n] = W.digits_batch(torch, row0 + s, n, 256, device=device)
        elif workload in ("c3", "c3s", "c3x", "c3s16", "c3x16", "c3u"):
            out[s:s + n] = W.keyword_batch(torch, words, row0 + s, n, 256, device=device)
        elif workload == "c3m16":
            out[s:s + n] = W.mixed_keyword_batch(torch, words, row0 + s, n, 256, device=device)
        elif workload == "c5w":
            out[s:s + n] = W.scriptseq_batch(torch, row0 + s, n, 256, device=device)
        else:
            out[s:s + n] = W.script_batch(torch, row0 + s, n, 256, device=device)
    return out


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(workload, pattern, rows_dev, op_name, budget_s):
    """Times the CPU oracle on a bounded sample of the same rows (rank 0, N = 1 only): all usable cores (OpenMP static
    over rows), then ONE core on a smaller sample (SURVEY.md s8d asks for both)."""
    import numpy as np
    from oracle.walker import Dfa, OraclePattern
    t = pattern.tables()
    d = {k: Dfa(t["class_map"], t["stride"], v["table"], v["accepting"], v["max_char"]) for k, v in t["dfas"].items()}
    o = OraclePattern(d["matches"], d["contained_in"], d["forwards"], d["backwards"], t["fixed_len"], -1)
    cores = usable_cores()
    fn = {"contained_in": o.batch_contained_in, "find": o.batch_find, "matches": o.batch_matches}[op_name]

    def timed(n, threads, budget):
        host = rows_dev[:n].cpu().numpy()
        if host.dtype == np.int16:
            host = host.view(np.uint16)
        fn(host[:4096], threads=threads)
        passes, t0 = 0, time.perf_counter()
        while True:
            fn(host, threads=threads)
            passes += 1
            el = time.perf_counter() - t0
            if el > budget or passes >= 64:
                break
        rows_s = passes * n / el
        return rows_s, rows_s * host.shape[1] * host.dtype.itemsize / 1e9, passes

    n_all = min(rows_dev.shape[0], 1 << 20)
    rows_s, gbs, passes = timed(n_all, cores, budget_s * 0.7)
    n_one = min(rows_dev.shape[0], 1 << 17)
    rows_s1, gbs1, passes1 = timed(n_one, 1, budget_s * 0.3)
    return {"value": gbs, "unit": "GB/s", "rows_per_s": rows_s, "cores": cores, "kind": "port",
            "sample": "%d passes over the first %d rows of the same batch (%s), OpenMP static over rows; "
                      "CPU restatement of the generated loops, not the JVM bytecode path" % (passes, n_all, workload),
            "single_core": {"value": gbs1, "unit": "GB/s", "rows_per_s": rows_s1, "cores": 1,
                            "sample": "%d passes over the first %d rows" % (passes1, n_one)}}


def measured_read_ceiling(buf):
    """This GPU's streaming-READ ceiling on the bench's own resident buffer: a trivial coalesced read-reduce kernel
    (needle_amd/csrc/stream_probe.hip, a measurement aid outside the product ABI), best of a few launch shapes with
    plain and with nontemporal loads."""
    import ctypes
    import torch
    from needle_amd.build import PROBE_LIB
    if not os.path.exists(PROBE_LIB):
        return None
    L = ctypes.CDLL(PROBE_LIB)
    L.stream_read_launch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    n = buf.numel() * buf.element_size()
    out = torch.zeros(4, dtype=torch.int32, device=buf.device)
    s = torch.cuda.current_stream().cuda_stream
    best = 0.0
    for blocks, unroll in ((2048, 1), (4096, 1), (2048, 4), (4096, 4), (4096, 8), (8192, 8), (2048, 104), (4096, 104), (4096, 108), (8192, 108)):  # 1xx = nt loads
        for _ in range(2):
            L.stream_read_launch(buf.data_ptr(), n, out.data_ptr(), blocks, unroll, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            L.stream_read_launch(buf.data_ptr(), n, out.data_ptr(), blocks, unroll, s)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, n * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return best


def must_read_bytes(workload, pattern, rows, cw):
    """Secondary, stricter denominator: sum over rows of the number of chars the reference loop
    touches before it stops, x bytes/char.  containedIn stops at the first accepting state = the END of the
    shortest-ending match prefix, which for C2 is the first digit (find().start + 1); unmatched rows are read whole.
    For find() (C3/C5) the forward walk runs to the char after `end` (where the search DFA dies) and the backward
    walk re-reads [start, end): an estimate, labelled as such."""
    import torch
    from needle_amd.pattern import unpack_bitmap
    n, L = rows.shape
    fw, fs, fe = pattern.find_batch(rows)
    m = torch.from_numpy(unpack_bitmap(fw, n)).to(rows.device)
    if workload == "c2":
        chars = torch.where(m, fs.long() + 1, torch.full_like(fs, L, dtype=torch.long))
        exact = True
    else:
        chars = torch.where(m, (fe.long() + 1).clamp(max=L) + (fe - fs).long(), torch.full_like(fs, L, dtype=torch.long))
        exact = False
    return int(chars.sum().item()) * cw, exact


class Ctx:
    pass


def verify_gather(sh, dev, row0, n_rows, total_rows, is_find, rank):
    """One extra step whose gathered results are checked against what every rank computed locally (see measure())."""
    import torch
    import torch.distributed as dist
    s = sh.step()
    full, st, en = sh.wait(s)
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize()

    def popcount(words):
        w = words.contiguous().view(torch.uint8)
        lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=w.device)
        return int(lut[w.long()].sum().item())

    def checksum(a, b, first_row):
        idx = (torch.arange(a.numel(), dtype=torch.int64, device=a.device) + first_row) % 65521 + 1
        return int(((a.long() + 3 * b.long() + 7) * idx).sum().item())

    local_words = (n_rows + 63) // 64
    if is_find and n_rows and "start" not in s:  # the scan stored the dword form itself (ShardedScan scan_packed): this rank's own halves
        loc_st, loc_en = sh._unpack(s["packed"][:n_rows].clone())
    elif is_find and n_rows:
        loc_st, loc_en = s["start"][:n_rows], s["end"][:n_rows]
    mine = torch.tensor([popcount(s["bitmap"][:local_words]) if n_rows else 0,
                         checksum(loc_st, loc_en, row0) if (is_find and n_rows) else 0],
                        dtype=torch.int64, device=dev)
    dist.all_reduce(mine)
    want_pop, want_sum = int(mine[0].item()), int(mine[1].item())
    res = {"ok": True, "popcount_ranks": want_pop}
    if full is not None:  # every rank (all-gather) or rank 0 (find's fan-in)
        got = popcount(full[:(total_rows + 63) // 64])
        res["popcount_gathered"] = got
        res["ok"] = res["ok"] and got == want_pop
        if is_find:
            got_sum = checksum(st[:total_rows], en[:total_rows], 0)
            res["checksum_ranks"], res["checksum_gathered"] = want_sum, got_sum
            res["ok"] = res["ok"] and got_sum == want_sum
    ok = torch.tensor([1 if res["ok"] else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    res["ok"] = bool(int(ok.item()))
    if not res["ok"] and rank == 0:
        sys.stderr.write("GATHER VERIFICATION FAILED: %r\n" % (res,))
    return res


def measure(workload, args, ctx, headline):
    """One workload, measured as the contract says: W warm-up steps, then exactly K steps bracketed by barrier +
    synchronize on both sides, max over ranks.  -> the dict that becomes the JSON line (headline) or an entry of
    "workloads"."""
    import torch
    import torch.distributed as dist
    from needle_amd.sharding import ShardedScan, shard_range
    from needle_amd.pattern import unpack_bitmap
    dev, world, rank, use_dist = ctx.dev, ctx.world, ctx.rank, ctx.use_dist
    pattern, what, words = make_pattern(workload)
    if headline and args.regex is not None:  # not a BASELINE config: labelled as such in config.workload
        from needle_amd.pattern import DFACompiler
        pattern, what = DFACompiler.compile(args.regex, "Custom"), "CUSTOM regex %r %s()" % (args.regex, args.op or "default op")
    total_rows = args.rows * world if args.scaling == "weak" else args.rows
    row0, n_rows = shard_range(total_rows, world, rank)
    rows = make_rows(workload, words, row0, n_rows, dev)
    cw = rows.element_size()
    op_name = (args.op if headline and args.op else None) or ("contained_in" if workload == "c2" else "find")
    op = {"contained_in": pattern.contained_in_batch, "find": pattern.find_batch, "matches": pattern.matches_batch}[op_name]
    is_find = op_name == "find"

    def scan(bitmap, start, end):
        op(rows, out=(bitmap, start, end) if is_find else bitmap)

    def scan_packed(bitmap, packed):  # needle_find_packed16_dev: the scan kernel stores the dword form itself
        pattern.find_packed16_batch(rows, out=(bitmap, packed))

    # N > 1, find: start / end cross the links as one dword per row (rows are 256 chars: two 16-bit halves), written by the scan
    # straight into the send buffer
    sh = ShardedScan(scan, total_rows, world, rank, is_find, dev, n_buffers=args.buffers, comm=ctx.comm, overlap=args.overlap == "on",
                     pack16=(use_dist or getattr(args, "force_pack16", False)) and is_find, max_row_len=rows.shape[1],
                     scan_packed=scan_packed if is_find else None)
    for _ in range(2):  # first launches: program upload, kernel attributes (never part of a captured graph)
        sh.scan_only()
    torch.cuda.synchronize()
    # Small shards (8 GPUs: 1.25M rows, a ~55 us scan): the scan is launched as a HIP graph so that the per-step host
    # cost is one graph launch instead of the library's argument marshalling + launch.
    graphs = None
    # (measured on 1.25M-row shards: graph replay 72 us per step against 63 us for plain launches -- the library's launch
    # path costs the host ~20 us, less than the 57 us scan it overlaps with, and a HIP graph launch is not cheaper.  So
    # the default is plain launches; --graph scan keeps the experiment reproducible.)
    want_graph = args.graph == "scan"
    if want_graph and n_rows:
        try:
            graphs = []
            for i in range(len(sh.sets)):
                sh.k = i
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    sh.scan_only()
                graphs.append(g)
            sh.k = 0
            sh.scan_only = lambda: (graphs[sh.k % len(graphs)].replay(), sh.sets[sh.k % len(sh.sets)])[1]
        except Exception as e:  # noqa: BLE001 -- capture is an optimisation; the eager path is always valid
            graphs = None
            sys.stderr.write("graph capture failed (%s): eager launches\n" % e)
            torch.cuda.synchronize()

    def step():
        # scan (bracketed by the two events) + the gathers, asynchronous on RCCL's stream and ordered after the kernel:
        # the next step's scan overlaps them.  Without a process group the gathers are no-ops.
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        return sh.step(ev), ev

    def fence():
        sh.drain()  # every step's results have landed (bitmap on every rank, start / end on rank 0)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Device pre-warm, untimed and not the kernel under test: a GPU that has been idle runs its first ~40 ms of load below
    # its steady clocks (measured on the 10M-row batch: C2 0.45 ms per step right after start-up, 0.404 from step ~100 on).
    # The W warm-up steps and the K timed steps below are the contract's; this only makes them steady-state steps.
    # The same K steps COLD first (device idle for 0.3 s, no pre-warm): what a caller sees who scans one batch now and then.
    cold = None
    if args.prewarm_ms > 0:
        fence()
        time.sleep(0.3)
        tc = time.perf_counter()
        cold_events = []
        for _ in range(args.steps):
            _l, ev = step()
            cold_events.append(ev)
        fence()
        cold_s = time.perf_counter() - tc
        if use_dist:
            tt = torch.tensor([cold_s], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            cold_s = float(tt.item())
        cold = (cold_s / args.steps, sum(a.elapsed_time(b) for a, b in cold_events) / len(cold_events))
    prewarm_ms = 0.0
    if args.prewarm_ms > 0 and n_rows:
        scratch = torch.empty_like(rows)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        while (time.perf_counter() - tp) * 1e3 < args.prewarm_ms:
            for _ in range(8):
                scratch.copy_(rows)
            torch.cuda.synchronize()
        prewarm_ms = (time.perf_counter() - tp) * 1e3
        del scratch
    for _ in range(args.warmup):
        last, _ev = step()
    fence()
    t0 = time.perf_counter()
    events = []
    for _ in range(args.steps):
        last, ev = step()
        events.append(ev)
    t_issue = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = sum(a.elapsed_time(b) for a, b in events) / len(events)
    # K timed steps are a few ms: a longer run right behind them (>= 100 ms of steps, same protocol) says whether the K steps
    # were representative of the steady state.  `value` stays the contract's exactly-K-steps figure.
    steady = None
    if n_rows or use_dist:
        ks = max(args.steps, int(0.1 / max(elapsed / args.steps, 1e-6)) + 1)
        if use_dist:
            kt = torch.tensor([ks], dtype=torch.int64, device=dev)
            dist.all_reduce(kt, op=dist.ReduceOp.MAX)
            ks = int(kt.item())
        t1 = time.perf_counter()
        sev = []
        for _ in range(ks):
            _l, ev = step()
            sev.append(ev)
        fence()
        steady_s = time.perf_counter() - t1
        if use_dist:
            tt = torch.tensor([steady_s], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            steady_s = float(tt.item())
        steady = (ks, steady_s / ks, sum(a.elapsed_time(b) for a, b in sev) / len(sev))
    # N > 1: what the gather delivered must be what the ranks computed -- a mis-ordered or truncated gather would otherwise
    # still print a clean line.  One more step, then: popcount of the gathered bitmap == sum over ranks of the shards'
    # popcounts, and for find() a position-weighted checksum of the gathered start / end == the sum of the ranks' own.
    gather_check = None
    if use_dist:
        gather_check = verify_gather(sh, dev, row0, n_rows, total_rows, is_find, rank)

    # algorithmic bytes: L*w input bytes + result bytes (1 bit/row; find adds 2 x int32/row)
    per_row = 256 * cw + (8 if is_find else 0)
    bytes_gpu = n_rows * per_row + ((n_rows + 63) // 64) * 8
    bytes_job = total_rows * per_row + ((total_rows + 63) // 64) * 8
    step_s = elapsed / args.steps
    matched = int(unpack_bitmap(last["bitmap"], n_rows).sum()) if n_rows else 0
    if use_dist:
        mt = torch.tensor([matched], dtype=torch.int64, device=dev)
        dist.all_reduce(mt)
        matched = int(mt.item())
    props = torch.cuda.get_device_properties(dev)
    inf = pattern.info()
    which = {"contained_in": "contained_in", "find": "forwards", "matches": "matches"}[op_name]
    # the kernel behind the op: the n-gram candidate filter kernel where the program carries a filter (8-bit rows, containedIn / find)
    # (UTF-16 rows of a pattern below 0xFF take the BYTE program's filter kernel, the text narrowed as it is loaded: the launches counted by
    # needle_pattern_prefilter_state say whether they did)
    utf16_route = cw == 2 and which != "matches" and inf["max_char"]["matches"] < 0xFF and pattern.prefilter_state(which)["filter_launches"] > 0
    pre = pattern.prefilter_info(which) if (cw == 1 or utf16_route) and which != "matches" else {"on": 0}
    # (UTF-16 rows of a pattern on several pages of the BMP: the WIDE filter -- windows of four code units)
    if cw == 2 and not utf16_route and which != "matches" and pattern.utf16_route() is None and pattern.prefilter_state(which)["filter_launches"] > 0:
        pre = pattern.prefilter_info(which, wide=True)
    kernel_name = "needle::ngram_kernel" if pre["on"] else "needle::scan_kernel"
    mode_names = {0: "packed functions", 1: "LDS table u8", 2: "LDS table u16", 3: "HBM table", 4: "LDS pair table", 5: "LDS hot rows + HBM table",
                  6: "compressed automaton in LDS (dense rows + exception records)"}
    out = {
        "value": bytes_job / step_s / 1e9,
        "unit": "GB/s",
        "ms_per_step": step_s * 1e3,
        "rows_per_s": total_rows / step_s,
        "matches_per_s": matched / step_s,
        "matched_fraction": matched / max(1, total_rows),
        "dtype": "u8" if cw == 1 else "u16",
        "config": {"workload": "%s: %s over %s%d x 256 %s rows%s" % (
                       workload, what, "" if world == 1 else "the SAME " if args.scaling == "strong" else "per-GPU ",
                       total_rows if args.scaling == "strong" else args.rows, "UTF-16" if cw == 2 else "ASCII",
                       "" if world == 1 else (", row-sharded over %d GPUs (%d rows per GPU)" % (world, n_rows))),
                   "rows_total": total_rows, "rows_per_gpu": n_rows, "row_chars": 256, "char_bytes": cw,
                   "parallelism": "row-shard x%d" % world,
                   "result": "bitmap" + ((("+start|end<<16, one dword per row (needle_find_packed16_dev)" if sh.scan_packed is not None else "+start/end int32") +
                                          (" (on the links: one dword per row, two 16-bit halves)" if use_dist else "")) if is_find else ""),
                   "automaton": {"states": inf["n_states"][which], "classes": inf["stride"],
                                 "kernel_mode": mode_names.get(inf["kernel_mode"][which], str(inf["kernel_mode"][which])),
                                 "prefilter": ({"windows": pre["n_windows"], "stride": pre["stride"], "run_up": pre["warm"], "bitmap_bytes": pre["bitmap_bytes"]}
                                               if pre["on"] else None)},
                   "launch": "HIP graph replay" if graphs else "eager"},
        "roofline": {"bound": "hbm", "achieved": bytes_gpu / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": bytes_gpu / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "kernel": kernel_name, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_gpu,
                     "chars_per_clk_per_cu": n_rows * 256 / (kernel_ms * 1e-3) / (ENGINE_CLOCK_MHZ * 1e6) / props.multi_processor_count},
        "host_issue_us_per_step": t_issue / args.steps * 1e6,
        "prewarm": {"ms": prewarm_ms, "what": "untimed copies of the batch before the W warm-up steps (steady clocks)"},
    }
    if cold is not None:
        out["cold"] = {"ms_per_step": cold[0] * 1e3, "kernel_ms": cold[1], "steps": args.steps,
                       "roofline.frac": bytes_gpu / (cold[1] * 1e-3) / 1e9 / HBM_PEAK_GBS if cold[1] > 0 else None,
                       "what": "the same K steps right after 0.3 s of device idle, before the pre-warm (first steps below steady clocks)"}
    if steady is not None:
        out["steady"] = {"steps_effective": steady[0], "ms_per_step": steady[1] * 1e3, "kernel_ms": steady[2],
                         "roofline.frac": bytes_gpu / (steady[2] * 1e-3) / 1e9 / HBM_PEAK_GBS if steady[2] > 0 else None,
                         "what": ">= 100 ms of steps right behind the K timed ones (same barriers; not the `value`)"}
    if gather_check is not None:
        out["gather_verified"] = gather_check["ok"]
        out["gather_check"] = gather_check
    # HBM traffic per launch comes from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected per
    # MI355X_MICROARCH.md) committed under profiles/ for this workload at this size -- and only if that profile was
    # taken with the kernels being benchmarked now (hash of the device sources); otherwise null.
    if args.rows == 10_000_000 and world == 1:
        import glob
        sha = kernel_source_sha()
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s.json" % workload)), reverse=True):
            try:
                prof = json.load(open(f))
                if prof.get("kernel_source_sha") == sha:
                    out["roofline"]["traffic"] = prof["traffic_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = os.path.relpath(f, ROOT)
                else:
                    out["roofline"]["traffic_note"] = "newest profile (%s) was taken with other kernel sources: not quoted" % os.path.relpath(f, ROOT)
                break
            except (OSError, KeyError, ValueError):
                pass
    if use_dist:
        # scan and gather reported separately (each blocking, nothing overlapped) beside the step time
        # in which the gathers ARE overlapped with the next step's scan
        def blocking(fn, reps=10):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / reps * 1e3
        out["scan_ms"] = blocking(lambda: sh.scan_only())
        out["gather_ms"] = blocking(lambda: sh.wait(sh.step())) - out["scan_ms"]
        out["step_ms"] = out["ms_per_step"]
        out["gather"] = {"collective": ("ONE gather to rank 0 (RCCL send/recv fan-in) of start / end as 16-bit halves | bitmap: %d B per rank" % (sh.sets[0]["buf"].numel() * 4)) if is_find
                                       else ("ONE all-gather (RCCL) of the bitmap words: %d B per rank" % (sh.per_words * 8)),
                         "issued_by": "libneedle_hip.so (needle_multi_*)" if ctx.comm is not None else "torch.distributed",
                         "note": "gather_ms = blocking (scan + gathers) - blocking scan; inside the timed steps the gathers overlap the next scan"}
    if use_dist and args.all_on_device is not None:
        # C4 rehearsal on ONE GPU (every rank's shard on the same device, gloo carrying the gathers): inside the timed steps the
        # ranks' kernels contend for the chip, so each rank also times its shard's scan ALONE, in turn -- what one GPU of an
        # 8-GPU node would spend on its 1/N of the batch.  N x the slowest of these against the N = 1 kernel time says what row
        # sharding itself costs (shorter launches, the tail) before any link is involved.
        solo = 0.0
        for r in range(world):
            dist.barrier()
            if r == rank and n_rows:
                for _ in range(5):
                    sh.scan_only()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(50):
                    sh.scan_only()
                b.record()
                torch.cuda.synchronize()
                solo = a.elapsed_time(b) / 50
        dist.barrier()
        tt = torch.tensor([solo], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        out["solo_kernel_ms"] = {"max_over_ranks": float(tt.item()), "x_ranks": float(tt.item()) * world,
                                 "what": "each rank's shard scanned alone on the shared device: 50 back-to-back launches between two HIP events; x_ranks = N x the "
                                         "slowest.  With N processes holding queues on one device this includes the driver's switching between "
                                         "their queues -- an upper bound on a rank's kernel, not the kernel (c4_shard_step has that)"}
    if headline and rank == 0 and world == 1 and not args.no_extras:
        out["roofline"]["device"] = {"name": props.name, "cus": props.multi_processor_count, "clock_mhz_nominal": ENGINE_CLOCK_MHZ}
        ceil = measured_read_ceiling(rows)
        if ceil:
            out["roofline"]["measured_read_ceiling"] = ceil
            out["roofline"]["frac_of_measured_ceiling"] = out["roofline"]["achieved"] / ceil
        mr, exact = must_read_bytes(workload if args.regex is None and args.op is None else "custom", pattern, rows, cw)
        out["must_read"] = {"bytes_per_step": mr, "GB/s": mr / step_s / 1e9, "exact": exact,
                            "note": "chars the reference loop touches before it stops x bytes/char (SURVEY.md s8d secondary denominator)"}
    if rank == 0 and world == 1 and not args.no_extras:
        # SURVEY.md s8d "results landed in host-visible memory": the same steps with the bitmap (and find()'s start /
        # end) copied to pinned host memory after every scan.  PCIe-bound for find (8 B per row); never the `value`.
        hb = torch.empty(sh.per_words, dtype=torch.int64).pin_memory()
        # (one pinned buffer of 8 B per row serves every find() variant below: start | end, the records, the packed dwords)
        hrec = torch.empty((n_rows, 2), dtype=torch.int32).pin_memory() if is_find else None
        hs = hrec.view(-1)[:n_rows] if is_find else None
        he = hrec.view(-1)[n_rows:] if is_find else None

        def landed():
            s = sh.scan_only()
            hb.copy_(s["bitmap"], non_blocking=True)
            if is_find:
                hs.copy_(s["start"][:n_rows], non_blocking=True)
                he.copy_(s["end"][:n_rows], non_blocking=True)
        k2 = max(3, args.steps // 4)

        def per_step(fn):
            """fn() k2 times, twice over; the better batch (a sporadic multi-ms stall of the copy engine / the host shows up in
            one batch of five steps now and then: seen on 1 of 6 figures per run, never twice in a row)."""
            fn()
            torch.cuda.synchronize()
            best = None
            for _ in range(2):
                t = time.perf_counter()
                for _ in range(k2):
                    fn()
                torch.cuda.synchronize()
                d = (time.perf_counter() - t) / k2
                best = d if best is None else min(best, d)
            return best
        dt = per_step(landed)
        out["host_landed"] = {"ms_per_step": dt * 1e3, "GB/s": bytes_job / dt / 1e9, "d2h_bytes_per_step": sh.per_words * 8 + (8 * n_rows if is_find else 0),
                              "note": "scan + D2H of the results into pinned host memory, every step"}
        if is_find and rows.shape[1] <= 65534:
            # the compact result forms (include/needle_hip.h): matched rows only as {row, start, end} records in row order
            # (needle_find_compact_dev: 8 B per MATCHED row over PCIe, after an 8-byte count), and start / end as one dword per row
            cw_, cr_, cc_ = torch.empty(sh.per_words, dtype=torch.int64, device=dev), torch.empty((n_rows, 2), dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
            hcnt = torch.zeros(1, dtype=torch.int64).pin_memory()
            d2h = [0]

            def landed_compact():
                pattern.find_compact(rows, out=(cw_, cr_, cc_))
                hcnt.copy_(cc_, non_blocking=True)
                hb.copy_(cw_, non_blocking=True)
                torch.cuda.current_stream().synchronize()  # the count decides how many records cross the bus
                k = int(hcnt.item())
                hrec[:k].copy_(cr_[:k], non_blocking=True)
                d2h[0] = sh.per_words * 8 + 8 + 8 * k
            dtc = per_step(landed_compact)
            out["host_landed"]["compact"] = {"ms_per_step": dtc * 1e3, "d2h_bytes_per_step": d2h[0], "matched_rows": int(hcnt.item()),
                                             "note": "needle_find_compact_dev + D2H of bitmap, count, then 8 B per MATCHED row"}
            pk = torch.empty(n_rows, dtype=torch.int32, device=dev)
            hpk = hs

            def landed_packed():
                pattern.find_packed16_batch(rows, out=(cw_, pk))
                hb.copy_(cw_, non_blocking=True)
                hpk.copy_(pk, non_blocking=True)
            dtp = per_step(landed_packed)
            # ... and that scan alone, device only (HIP events): the same kernel as the headline's with 4 result bytes per row instead of 8
            pev = []
            for _ in range(max(3, args.steps)):
                a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_.record()
                pattern.find_packed16_batch(rows, out=(cw_, pk))
                b_.record()
                pev.append((a_, b_))
            torch.cuda.synchronize()
            out["host_landed"]["packed16"] = {"ms_per_step": dtp * 1e3, "d2h_bytes_per_step": sh.per_words * 8 + 4 * n_rows,
                                              "scan_kernel_ms": sum(x.elapsed_time(y) for x, y in pev) / len(pev),
                                              "note": "needle_find_packed16_dev (the scan stores one dword per row itself) + D2H of 4 B per row; "
                                                      "scan_kernel_ms: that scan alone on the device"}
            del cw_, cr_, cc_, pk
    if rank == 0 and world == 1 and not args.no_extras:
        # ... and PIPELINED ("results landed in host-visible memory" as a steady-state rate): two result sets; step k's D2H
        # runs on a copy stream under step k + 1's scan.  find(): the one-dword form the scan stores itself (4 B per row over PCIe).
        n_sets = 2
        s_scan, s_copy = torch.cuda.current_stream(), torch.cuda.Stream()

        def pipelined(form):
            """form: None (containedIn / int32 arrays), "packed16" (one dword per row), "packed8" (one uint16 per row: rows <= 256 chars)"""
            dsets, hsets = [], []
            for _ in range(n_sets):
                d = {"bitmap": torch.empty(sh.per_words, dtype=torch.int64, device=dev)}
                h = {"bitmap": torch.empty(sh.per_words, dtype=torch.int64).pin_memory()}
                if is_find and form == "packed16":
                    d["packed"] = torch.empty(n_rows, dtype=torch.int32, device=dev)
                    h["packed"] = torch.empty(n_rows, dtype=torch.int32).pin_memory()
                elif is_find and form == "packed8":
                    d["packed"] = torch.empty(n_rows, dtype=torch.int16, device=dev)
                    h["packed"] = torch.empty(n_rows, dtype=torch.int16).pin_memory()
                elif is_find:
                    d["start"], d["end"] = torch.empty(n_rows, dtype=torch.int32, device=dev), torch.empty(n_rows, dtype=torch.int32, device=dev)
                    h["start"], h["end"] = torch.empty(n_rows, dtype=torch.int32).pin_memory(), torch.empty(n_rows, dtype=torch.int32).pin_memory()
                dsets.append(d)
                hsets.append(h)
            scan_done = [torch.cuda.Event() for _ in range(n_sets)]
            copy_done = [torch.cuda.Event() for _ in range(n_sets)]

            def pipelined_steps(k_steps):
                for i in range(k_steps):
                    k = i % n_sets
                    s_scan.wait_event(copy_done[k])  # set k's previous results have left the device
                    d = dsets[k]
                    if not is_find:
                        op(rows, out=d["bitmap"])
                    elif form == "packed16":
                        pattern.find_packed16_batch(rows, out=(d["bitmap"], d["packed"]))
                    elif form == "packed8":
                        pattern.find_packed8_batch(rows, out=(d["bitmap"], d["packed"]))
                    else:
                        op(rows, out=(d["bitmap"], d["start"], d["end"]))
                    scan_done[k].record(s_scan)
                    s_copy.wait_event(scan_done[k])
                    with torch.cuda.stream(s_copy):
                        for key, t in d.items():
                            hsets[k][key].copy_(t, non_blocking=True)
                        copy_done[k].record(s_copy)
                torch.cuda.synchronize()
            for e_ in copy_done:
                e_.record(s_copy)
            pipelined_steps(4)
            kp = max(8, args.steps)
            best = None
            for _ in range(2):
                tq = time.perf_counter()
                pipelined_steps(kp)
                dq = (time.perf_counter() - tq) / kp
                best = dq if best is None else min(best, dq)
            if form == "packed8":  # the 2-byte form against the dword form on the last set scanned (same rows)
                import numpy as np
                w16, p16 = pattern.find_packed16_batch(rows)
                from needle_amd.pattern import Pattern
                s8, e8 = Pattern.unpack8(dsets[(kp - 1) % n_sets]["packed"][:200000].cpu().numpy())
                v16 = p16[:200000].cpu().numpy().view(np.uint32).astype(np.int64)
                no = v16 == 0xFFFFFFFF
                assert ((s8 == np.where(no, -1, v16 & 0xFFFF)) & (e8 == np.where(no, -1, v16 >> 16))).all()
            per_row = {None: 8, "packed16": 4, "packed8": 2}[form] if is_find else 0
            d2h = sh.per_words * 8 + per_row * n_rows
            return {"ms_per_step": best * 1e3, "GB/s": bytes_job / best / 1e9, "frac_of_hbm_peak": bytes_job / best / 1e9 / HBM_PEAK_GBS,
                    "d2h_bytes_per_step": d2h, "d2h_GB/s": d2h / best / 1e9}

        form = None if not is_find else ("packed8" if rows.shape[1] <= 256 else "packed16" if rows.shape[1] <= 65534 else None)
        out["host_landed"]["pipelined"] = pipelined(form)
        out["host_landed"]["pipelined"]["form"] = form or ("bitmap" if not is_find else "int32 arrays")
        out["host_landed"]["pipelined"]["note"] = ("steady state of scan k + 1 on the launch stream beside the D2H of step k's results on a copy stream (two result sets); "
                                                   "find(): the result form named in `form` -- packed8 = one uint16 per row (needle_find_packed8_dev: rows <= 256 chars), "
                                                   "packed16 = one dword per row")
        if form == "packed8":
            out["host_landed"]["pipelined16"] = pipelined("packed16")  # round 5's figure: 4 result bytes per row over PCIe
    if rank == 0 and world == 1 and not args.no_extras and workload in ("c2", "c3", "c3s", "c3x", "c5", "c3s16", "c3x16", "c3m16"):
        # EVERY non-overlapping match of every row (the reference's repeated find()), one pass over
        # the batch (needle_find_all.hip; for dictionaries behind the n-gram candidate filter -- c3s -- that kernel's find-all form,
        # needle_ngram.hip): counts + dense per-row slots.  Its own figure, never the `value`.
        slots = 32
        fc = torch.zeros(n_rows, dtype=torch.int32, device=dev)
        fs = torch.full((n_rows, slots), -1, dtype=torch.int32, device=dev)
        fe = torch.full((n_rows, slots), -1, dtype=torch.int32, device=dev)
        more = pattern.find_all_dense(rows, slots, out=(fc, fs, fe))[3]
        torch.cuda.synchronize()
        k2 = max(3, args.steps // 4)
        t = time.perf_counter()
        for _ in range(k2):
            pattern.find_all_dense(rows, slots, out=(fc, fs, fe))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / k2
        n_matches = int(fc.sum().item())
        fa_bytes = n_rows * (256 * cw + 4) + 8 * n_matches
        out["find_all"] = {"ms_per_step": dt * 1e3, "matches": n_matches, "matches_per_s": n_matches / dt, "max_per_row": int(fc.max().item()),
                           "slots": slots, "more": bool(more), "GB/s": fa_bytes / dt / 1e9, "algorithmic_bytes": fa_bytes,
                           "kernel": ("needle::ngram_kernel (find-all form)" if (pre["on"] and is_find) else
                                      "needle::find_all_lockstep_kernel" if pattern.find_all_transducer(cw) is not None else "needle::find_all_kernel"),
                           "frac_of_hbm_peak": fa_bytes / dt / 1e9 / HBM_PEAK_GBS,
                           "note": "every non-overlapping match per row (repeated Matcher.find()), one pass; bytes = rows + 4 B count per row + 8 B per match"}
        # the same with each match as one dword (needle_find_all_packed16_dev: start | end << 16): one result line per row
        t = time.perf_counter()
        for _ in range(k2):
            pattern.find_all_dense_packed16(rows, slots, out=(fc, fs))
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - t) / k2
        assert int(fc.sum().item()) == n_matches and ((fs >> 16) & 0xFFFF)[:, 0][fc > 0].eq(fe[:, 0][fc > 0]).all()
        if workload == "c2":  # '[0-9]+': every row's first match is what find() reports -- and the batch's rows have at most one run of digits
            fb_, fs_, fe_ = pattern.find_batch(rows)
            assert fe_[fc > 0].eq(fe[:, 0][fc > 0]).all() and fs_[fc > 0].eq(fs[:, 0][fc > 0] & 0xFFFF).all()
            del fb_, fs_, fe_
        out["find_all"]["packed16"] = {"ms_per_step": dtp * 1e3, "matches_per_s": n_matches / dtp,
                                       "algorithmic_bytes": n_rows * (256 * cw + 4) + 4 * n_matches,
                                       "frac_of_hbm_peak": (n_rows * (256 * cw + 4) + 4 * n_matches) / dtp / 1e9 / HBM_PEAK_GBS}
        # one dword per match in GROUP-BLOCKED slots (needle_find_all_blocked16_dev: slot k of 64 consecutive rows is one 256-byte run --
        # the lanes of a group fill the same lines): the same matches, checked against the row-major form
        fb = torch.full(((n_rows + 63) // 64, slots, 64), -1, dtype=torch.int32, device=dev)
        pattern.find_all_blocked16(rows, slots, out=(fc, fb), want_more=False)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(k2):
            pattern.find_all_blocked16(rows, slots, out=(fc, fb), want_more=False)
        torch.cuda.synchronize()
        dtb = (time.perf_counter() - t) / k2
        assert int(fc.sum().item()) == n_matches
        for k in range(2):  # slots 0 and 1 of every row against the row-major form
            assert fb[:, k, :].reshape(-1)[:n_rows][fc > k].eq(fs[:, k][fc > k]).all()
        out["find_all"]["blocked16"] = {"ms_per_step": dtb * 1e3, "matches_per_s": n_matches / dtb,
                                        "algorithmic_bytes": n_rows * (256 * cw + 4) + 4 * n_matches,
                                        "frac_of_hbm_peak": (n_rows * (256 * cw + 4) + 4 * n_matches) / dtb / 1e9 / HBM_PEAK_GBS}
        del fb
        # the counting pass alone (needle_count_matches_dev): the walk without filing
        t = time.perf_counter()
        for _ in range(k2):
            pattern.count_matches_batch(rows)
        torch.cuda.synchronize()
        out["find_all"]["count_pass_ms"] = (time.perf_counter() - t) / k2 * 1e3
        del fc, fs, fe
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(workload, pattern, rows, op_name, 10.0 if headline else 4.0)
    del rows, sh, graphs
    torch.cuda.empty_cache()
    return out, total_rows


def measure_ragged(workload, args, ctx):
    """The ragged variant SURVEY.md s8(d) / BASELINE.md s2 name: the same 10^7 x 256 batch with per-row lengths uniform in [1, 256]
    (a length table beside the rows; the kernels' GUARD instantiations).  The roofline fraction is over the ACTUAL bytes -- the chars
    inside the rows' lengths + 4 B of length + the result bytes per row -- not the nominal 256 per row.  c2r: '[0-9]+' containedIn(),
    c3r: the 1000-keyword union find().  W warm-up steps, K steps between two HIP events on the launch stream."""
    import torch
    from needle_amd.pattern import unpack_bitmap
    base = workload[:-1]
    pattern, what, words = make_pattern(base)
    n = args.rows
    rows = make_rows(base, words, 0, n, ctx.dev)
    lens = (torch.arange(n, device=ctx.dev, dtype=torch.int64) * 2654435761 % 256 + 1).to(torch.int32)
    is_find = base != "c2"
    words_n = (n + 63) // 64
    bitmap = torch.empty(words_n, dtype=torch.int64, device=ctx.dev)
    st = torch.empty(n, dtype=torch.int32, device=ctx.dev) if is_find else None
    en = torch.empty(n, dtype=torch.int32, device=ctx.dev) if is_find else None

    def run():
        if is_find:
            pattern.find_batch(rows, lens, out=(bitmap, st, en))
        else:
            pattern.contained_in_batch(rows, lens, out=bitmap)
    for _ in range(max(2, args.warmup)):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    chars = int(lens.sum().item())
    actual = chars + n * (4 + (8 if is_find else 0)) + words_n * 8
    matched = int(unpack_bitmap(bitmap, n).sum())
    out = {"workload": "%s: %s, per-row lengths uniform in [1, 256] over %d x 256 ASCII rows" % (workload, what, n), "ms_per_step": ms,
           "actual_bytes": actual, "chars": chars, "GB/s": actual / ms / 1e6, "frac_of_hbm_peak": actual / ms / 1e6 / HBM_PEAK_GBS,
           "nominal_GB/s": n * 256 / ms / 1e6, "matched_fraction": matched / n,
           "note": "bytes = chars inside the rows' lengths + 4 B length + result bytes per row; the rows stay 256 bytes apart, so the lines a "
                   "short row shares with nothing else are fetched all the same (nominal_GB/s = 256 B per row / time)"}
    if not is_find:  # '[0-9]+': recomputed with torch on the device
        col = torch.arange(256, device=ctx.dev, dtype=torch.int32)[None, :]
        want = 0
        for s0 in range(0, n, 1 << 20):
            r = rows[s0:s0 + (1 << 20)]
            want += int((((r >= 48) & (r <= 57)) & (col < lens[s0:s0 + (1 << 20), None])).any(dim=1).sum().item())
        assert want == matched, ("c2r containedIn", want, matched)
        out["verified"] = "torch recomputation of all %d rows" % n
    del rows, lens, bitmap
    return out


def _r(x, n=4):
    return round(x, n) if isinstance(x, float) else x


def slim(w):
    """One workload's digest for the stdout line: step time, roofline of its kernel, traffic / algorithmic bytes, host-landed and
    find-all figures, the CPU baseline -- numbers only (the notes are in the full JSON on stderr / bench_full.json)."""
    if "error" in w:
        return w
    r = w["roofline"]
    o = {"ms": _r(w["ms_per_step"]), "kernel_ms": _r(r["kernel_ms"]), "GBs": _r(r["achieved"], 0), "frac": _r(r["frac"], 3),
         "kernel": r["kernel"].replace("needle::", "").replace("_kernel", ""), "mode": w["config"]["automaton"]["kernel_mode"][:14],
         "states": w["config"]["automaton"]["states"],
         "traffic_x": _r(r["traffic"] / r["algorithmic_bytes_per_launch"], 2) if r.get("traffic") else None,
         "match_s": _r(w["matches_per_s"], 0), "matched": _r(w["matched_fraction"], 3)}
    if "cold" in w:
        o["cold_ms"], o["steady_ms"] = _r(w["cold"]["ms_per_step"], 3), _r(w["steady"]["ms_per_step"], 3) if "steady" in w else None
    if "gather_verified" in w:
        o["gather_verified"], o["scan_ms"], o["gather_ms"] = w["gather_verified"], _r(w.get("scan_ms")), _r(w.get("gather_ms"))
    hl = w.get("host_landed")
    if hl and "pipelined" in hl:  # results landed in host memory, steady state (scan k + 1 beside the D2H of step k)
        o["host_ms"] = {"pipelined": _r(hl["pipelined"]["ms_per_step"], 3), "frac": _r(hl["pipelined"]["frac_of_hbm_peak"], 3), "form": hl["pipelined"].get("form")}
        if "pipelined16" in hl:
            o["host_ms"]["dword"] = _r(hl["pipelined16"]["ms_per_step"], 3)
    fa = w.get("find_all")
    if fa:  # every match of every row: two arrays / one dword per match row-major / one dword per match in group-blocked slots / count only
        o["find_all"] = {"ms": _r(fa["ms_per_step"], 3), "dword_ms": _r(fa["packed16"]["ms_per_step"], 3), "blocked_ms": _r(fa["blocked16"]["ms_per_step"], 3),
                         "blocked_frac": _r(fa["blocked16"]["frac_of_hbm_peak"], 3), "count_ms": _r(fa.get("count_pass_ms"), 3),
                         "Gmatch_s": _r(fa["blocked16"]["matches_per_s"] / 1e9, 2), "kernel": fa["kernel"].replace("needle::", "").replace("_kernel", "")[:18]}
    cb = w.get("cpu_baseline")
    if cb:
        o["cpu_GBs"] = {"all": _r(cb["value"], 2), "cores": cb["cores"], "one": _r(cb["single_core"]["value"], 2)}
    return o


def slim_line(out):
    """The stdout line: the contract's fields, `roofline` and `cpu_baseline` as the contract asks for them, then digests."""
    keep = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"]
    line = {k: out[k] for k in keep}
    line["config"] = {k: v for k, v in out["config"].items() if k != "automaton"}
    line["config"]["automaton"] = {k: out["config"]["automaton"][k] for k in ("states", "classes", "kernel_mode")}
    line["roofline"] = {k: (_r(v, 6) if isinstance(v, float) else v) for k, v in out["roofline"].items() if k not in ("device", "traffic_note")}
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = {"value": _r(cb["value"], 3), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"][:110],
                                "single_core_GBs": _r(cb["single_core"]["value"], 3)}
    for k in ("rows_per_s", "matches_per_s", "gather_verified", "scan_ms", "gather_ms", "kernel_source_sha"):
        if k in out:
            line[k] = _r(out[k], 1) if isinstance(out[k], float) else out[k]
    if "cold" in out:
        line["cold_ms"], line["steady_ms"] = _r(out["cold"]["ms_per_step"]), _r(out["steady"]["ms_per_step"]) if "steady" in out else None
    if "must_read" in out:
        line["must_read_GBs"] = _r(out["must_read"]["GB/s"], 1)
    hl = out.get("host_landed")
    if hl and "pipelined" in hl:
        line["host_ms"] = {"plain": _r(hl["ms_per_step"], 3), "pipelined": _r(hl["pipelined"]["ms_per_step"], 3), "frac": _r(hl["pipelined"]["frac_of_hbm_peak"], 3),
                           "form": hl["pipelined"].get("form")}
    if "find_all" in out:
        line["find_all"] = slim({"roofline": out["roofline"], "ms_per_step": out["ms_per_step"], "config": out["config"], "matches_per_s": out["matches_per_s"],
                                 "matched_fraction": out["matched_fraction"], "find_all": out["find_all"]})["find_all"]
    if "c4_shard_step" in out:
        line["c4_shard_step"] = {w: ({"ms": _r(v["ms_per_step"]), "kernel_ms": _r(v["kernel_ms"]), "overhead": _r(v["overhead_frac"], 3)} if "error" not in v else v)
                                 for w, v in out["c4_shard_step"].items()}
    if "workloads" in out:
        line["workloads"] = {w: slim(v) for w, v in out["workloads"].items()}
    if "ragged" in out:
        line["ragged"] = {w: ({"ms": _r(v["ms_per_step"]), "GBs": _r(v["GB/s"], 1), "frac": _r(v["frac_of_hbm_peak"]), "nominal_GBs": _r(v["nominal_GB/s"], 1)}
                              if "error" not in v else v) for w, v in out["ragged"].items()}
    line["full"] = "stderr, bench_full.json"
    return line


def error_line(args, msg, emit=True):
    """A run that cannot start still prints ONE JSON line (the contract's keys, `value` null, the reason under "error")."""
    if emit:
        print(json.dumps({"metric": "GB/s haystack scanned (10M x 256-char batch, DFA table walk)", "value": None, "unit": "GB/s", "n_gpus": args.gpus,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling,
                          "vs_baseline": None, "dtype": "u8", "data": "synthetic", "config": {"workload": args.workload}, "error": msg}), flush=True)


def self_launch(args, torch):
    """`python bench.py --gpus N` (N > 1) outside a launcher: re-run this command line as N ranks under torch.distributed.run on
    127.0.0.1 with a free port; the children inherit stdout / stderr, so rank 0's line is this process's line.  Returns the exit code."""
    import socket
    import subprocess
    if args.all_on_device is None and torch.cuda.device_count() < args.gpus:
        error_line(args, "--gpus %d but %d device(s) visible" % (args.gpus, torch.cuda.device_count()))
        return 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("MASTER_PORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c3s", "c3x", "c5", "c5w", "c3s16", "c3x16", "c3m16", "c3u"], help="the headline workload")
    ap.add_argument("--also", default=None, help="comma list of further workloads measured into \"workloads\" "
                    "(default: c3,c3s,c3x,c5,c5w,c3s16,c3x16 at 1 GPU, c3 at N > 1; 'none' for profiling runs; c3s16 / c3x16: the c3s / c3x dictionaries over UTF-16 rows)")
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows in total (strong) or per GPU (weak)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"])
    ap.add_argument("--graph", default="off", choices=["off", "scan"], help="launch the scan as a HIP graph (experiment; plain launches are faster)")
    ap.add_argument("--buffers", type=int, default=2, help="result buffer sets rotating through the steps (N > 1: gathers in flight)")
    ap.add_argument("--overlap", default="off", choices=["on", "off"], help="N > 1: gather on a side stream beside the next scan (costs ~25 us of event handshakes per step)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (gloo + --all-on-device 0: a functional N > 1 run on ONE GPU; tests)")
    ap.add_argument("--all-on-device", type=int, default=None, help="tests: every rank uses this device instead of LOCAL_RANK")
    ap.add_argument("--collectives", default="rccl", choices=["rccl", "torch"], help="N > 1: gathers through the library's RCCL communicator (default) or torch.distributed")
    ap.add_argument("--regex", default=None, help="tuning runs: another regex over the chosen workload's rows")
    ap.add_argument("--op", default=None, choices=["matches", "contained_in", "find"], help="tuning runs: another op")
    ap.add_argument("--prewarm-ms", type=float, default=150.0, help="untimed device pre-warm (plain copies of the batch) before the W warm-up steps: "
                    "an idle GPU runs its first ~40 ms of load below its steady clocks; 0 = off")
    ap.add_argument("--full-line", action="store_true", help="print everything measured on stdout (one long JSON line) instead of the digest; "
                    "the default keeps the line under the 8 KB the driver's log tail holds and sends the full JSON to stderr / bench_full.json")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the read-ceiling probe, the must-read byte count and the host-landed figure")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver (and README.md) call it at N = 1: start the N ranks ourselves, one process
        # per GPU, and let rank 0 print the ONE line.  Under torch.distributed.run (RANK / WORLD_SIZE set) this is skipped.
        raise SystemExit(self_launch(args, torch))
    ctx = Ctx()
    ctx.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    ctx.rank = rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        error_line(args, "--gpus %d but WORLD_SIZE=%d: the launcher's --nproc-per-node must equal --gpus" % (args.gpus, world), rank == 0)
        raise SystemExit(2)
    if args.all_on_device is None and torch.cuda.device_count() < max(args.gpus, 1):
        error_line(args, "--gpus %d but %d device(s) visible" % (args.gpus, torch.cuda.device_count()), rank == 0)
        raise SystemExit(3)
    if args.all_on_device is not None:
        local = args.all_on_device
    torch.cuda.set_device(local)
    ctx.dev = dev = torch.device("cuda", local)
    ctx.use_dist = use_dist = world > 1 or "RANK" in os.environ  # also under `torch.distributed.run --nproc-per-node 1`
    if use_dist and args.overlap == "on":
        # the scan kernel is one persistent workgroup per CU that owns the CU's whole LDS; leave a few CUs free so
        # that the RCCL kernels gathering the PREVIOUS step's results can run next to it instead of behind it
        os.environ.setdefault("NEEDLE_RESERVE_CUS", "4")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            args.collectives = "torch"  # (the library's communicator is RCCL)

    ctx.comm = None
    if use_dist and args.collectives == "rccl":
        # the gathers go through the library's own RCCL communicator (one C call per step); torch.distributed only
        # carries the communicator id, the barrier and the max-over-ranks reduction of the clock
        from needle_amd.multi import RankComm
        try:
            ctx.comm = RankComm.from_torch_distributed(dev)
        except Exception as e:  # noqa: BLE001 -- e.g. no loadable RCCL outside torch's: the torch.distributed path is equivalent
            sys.stderr.write("library RCCL communicator unavailable (%s): gathers through torch.distributed\n" % e)
            ctx.comm = None
        ok = torch.tensor([1 if ctx.comm is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # every rank must take the same path
        if int(ok.item()) == 0:
            ctx.comm = None
    if args.also is None:
        also = ["c3", "c3s", "c3x", "c5", "c5w", "c3s16", "c3x16", "c3m16", "c3u"] if world == 1 else ["c3"]
        if args.regex or args.op or args.rows != 10_000_000:
            also = []
    else:
        also = [w for w in args.also.split(",") if w and w != "none"]
    also = [w for w in also if w != args.workload]

    head, total_rows = measure(args.workload, args, ctx, True)
    out = {
        "metric": "GB/s haystack scanned (10M x 256-char batch, DFA table walk)",
        "value": head.pop("value"),
        "unit": head.pop("unit"),
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": head.pop("ms_per_step"),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": head.pop("dtype"),
        "data": "synthetic",
    }
    out.update(head)
    out["kernel_source_sha"] = kernel_source_sha()
    if world == 1 and not use_dist and not args.no_extras and args.rows == 10_000_000 and args.regex is None and args.op is None:
        # C4's per-GPU share at 8 GPUs, measured on this one GPU: the step over a 1.25M-row shard of the
        # headline batch (and of C3's, whose find() gathers start / end) -- scan kernel vs whole step, i.e. what launch
        # overhead costs at that size.  No communication here; N > 1 runs report scan_ms / gather_ms themselves.
        import copy
        small = copy.copy(args)
        small.rows, small.steps, small.warmup, small.no_extras, small.no_cpu_baseline = 1_250_000, 200, 10, True, True
        small.force_pack16 = True  # find(): the form a shard's scan stores on an N > 1 run -- one dword per row, written by the kernel itself
        out["c4_shard_step"] = {}
        for w in [args.workload] + [x for x in ("c3",) if x != args.workload]:
            try:
                r, _ = measure(w, small, ctx, False)
                out["c4_shard_step"][w] = {"rows": small.rows, "ms_per_step": r["ms_per_step"], "kernel_ms": r["roofline"]["kernel_ms"],
                                           "overhead_frac": r["ms_per_step"] / r["roofline"]["kernel_ms"] - 1.0,
                                           "result": r["config"]["result"],
                                           "host_issue_us_per_step": r["host_issue_us_per_step"]}
            except Exception as e:  # noqa: BLE001
                out["c4_shard_step"][w] = {"error": "%s: %s" % (type(e).__name__, e)}
    if also:
        out["workloads"] = {}
        for w in also:
            try:
                r, _ = measure(w, args, ctx, False)
            except Exception as e:  # noqa: BLE001 -- a further workload must never cost the headline its line
                r = {"error": "%s: %s" % (type(e).__name__, e)}
            out["workloads"][w] = r
    if world == 1 and not use_dist and not args.no_extras and args.rows == 10_000_000 and args.regex is None and args.op is None and args.also is None:
        out["ragged"] = {}
        for w in ("c2r", "c3r"):
            try:
                out["ragged"][w] = measure_ragged(w, args, ctx)
            except Exception as e:  # noqa: BLE001
                out["ragged"][w] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        # The driver keeps the last 8 KB of stdout: the ONE line printed there is the contract's fields + roofline + cpu_baseline +
        # a compact digest of every workload (slim()); everything measured, with its notes, goes to stderr and bench_full.json.
        full = json.dumps(out)
        sys.stderr.write(full + "\n")
        try:
            with open(os.path.join("gpurun_out" if os.path.isdir("gpurun_out") else ".", "bench_full.json"), "w") as fh:
                fh.write(full + "\n")
        except OSError:
            pass
        print(full if args.full_line else json.dumps(slim_line(out)))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
