#!/usr/bin/env python3
"""bench.py -- the BASELINE.json metric on MI355X: GB/s of haystack scanned (+ rows/s) by the DFA table-walk
hot path on the 10M x 256-char synthetic batch.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c5] [--rows R] [--scaling weak|strong]

One "step" = one pass of the hot path over the whole device-resident batch (one kernel launch per GPU; for
N > 1 followed by the RCCL gather of the result bitmap to rank 0).  N > 1 is launched by the driver with
torch.distributed.run, one rank per GPU; rows are sharded by contiguous row blocks, no data-path collective
other than the result gather.  Rank 0 prints ONE JSON line.

  value       whole-job algorithmic GB/s (input bytes + result bytes, per step, all GPUs) over
              the barrier-bracketed wall time of exactly K steps (max over ranks), inputs resident in HBM
  roofline    the scan kernel alone: algorithmic bytes per launch / mean launch duration from HIP events
              recorded on the launch stream inside the timed region; peak = 8 TB/s HBM3E
  cpu_baseline  the CPU oracle (oracle/needle_walk.c, a port of the reference's generated loops -- NOT the JVM
              bytecode path: no JDK on the box) on a bounded sample of the same rows, all host cores
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL across processes)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
ENGINE_CLOCK_MHZ = 2400.0  # MI355X peak engine clock (same guide); chars/clk/CU is quoted against it


def make_pattern(workload):
    from needle_amd import workload as W
    from needle_amd.pattern import DFACompiler, LEFTMOST_LONGEST  # noqa: F401
    if workload == "c2":
        return DFACompiler.compile("[0-9]+", "DigitPlus"), "'[0-9]+' containedIn()", None
    if workload == "c3":
        words = W.keywords(1000)
        return DFACompiler.compile("|".join(words), "Keywords1k"), "union-of-1k-keywords find()", words
    if workload == "c5":
        return DFACompiler.compile(W.script_regex(), "ScriptRuns"), "BMP char-class regex find() over UTF-16", None
    raise SystemExit("unknown workload " + workload)


def make_rows(workload, words, row0, n_rows, device):
    """Shard [row0, row0 + n_rows) of the synthetic batch, generated on the GPU in slabs."""
    import torch
    from needle_amd import workload as W
    dtype = torch.int16 if workload == "c5" else torch.uint8
    out = torch.empty((n_rows, 256), dtype=dtype, device=device)
    slab = 1 << 19
    for s in range(0, n_rows, slab):
        n = min(slab, n_rows - s)
        if workload == "c2":
            out[s:s + n] = W.digits_batch(torch, row0 + s, n, 256, device=device)
        elif workload == "c3":
            out[s:s + n] = W.keyword_batch(torch, words, row0 + s, n, 256, device=device)
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


def cpu_baseline(workload, pattern, rows_dev, op_name, budget_s=12.0):
    """Times the CPU oracle on a bounded sample of the same rows (rank 0, N = 1 only)."""
    import numpy as np
    from oracle.walker import Dfa, OraclePattern
    t = pattern.tables()
    d = {k: Dfa(t["class_map"], t["stride"], v["table"], v["accepting"], v["max_char"]) for k, v in t["dfas"].items()}
    o = OraclePattern(d["matches"], d["contained_in"], d["forwards"], d["backwards"], t["fixed_len"], -1)
    cores = usable_cores()
    n = min(rows_dev.shape[0], 1 << 20)
    host = rows_dev[:n].cpu().numpy()
    if host.dtype == np.int16:
        host = host.view(np.uint16)
    fn = {"contained_in": o.batch_contained_in, "find": o.batch_find, "matches": o.batch_matches}[op_name]
    fn(host[:4096], threads=cores)
    passes, t0 = 0, time.perf_counter()
    while True:
        fn(host, threads=cores)
        passes += 1
        el = time.perf_counter() - t0
        if el > budget_s or passes >= 64:
            break
    rows_s = passes * n / el
    in_bytes = host.shape[1] * host.dtype.itemsize
    return {"value": rows_s * in_bytes / 1e9, "unit": "GB/s", "rows_per_s": rows_s, "cores": cores, "kind": "port",
            "sample": "%d passes over the first %d rows of the same batch (%s), OpenMP static over rows; "
                      "CPU restatement of the generated loops, not the JVM bytecode path" % (passes, n, workload)}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5"])
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per GPU (weak) or in total (strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--regex", default=None, help="tuning runs: another regex over the chosen workload's rows")
    ap.add_argument("--op", default=None, choices=["matches", "contained_in", "find"], help="tuning runs: another op")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the read-ceiling probe and the must-read byte count")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N > 1 through torch.distributed.run" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or "RANK" in os.environ  # also under `torch.distributed.run --nproc-per-node 1`
    if world > 1:
        # the scan kernel is one persistent workgroup per CU that owns the CU's whole LDS; leave a few CUs free so
        # that the RCCL kernel gathering the PREVIOUS step's bitmap can run next to it instead of behind it
        os.environ.setdefault("NEEDLE_RESERVE_CUS", "4")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from needle_amd.sharding import shard_range, gather_bitmap_async
    pattern, what, words = make_pattern(args.workload)
    if args.regex is not None:  # not a BASELINE config: labelled as such in config.workload
        from needle_amd.pattern import DFACompiler
        pattern, what = DFACompiler.compile(args.regex, "Custom"), "CUSTOM regex %r %s()" % (args.regex, args.op or "default op")
    total_rows = args.rows * world if args.scaling == "weak" else args.rows
    row0, n_rows = shard_range(total_rows, world, rank)
    rows = make_rows(args.workload, words, row0, n_rows, dev)
    cw = rows.element_size()
    op_name = args.op or ("contained_in" if args.workload == "c2" else "find")
    op = {"contained_in": pattern.contained_in_batch, "find": pattern.find_batch, "matches": pattern.matches_batch}[op_name]
    is_find = op_name == "find"

    def step():
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        res = op(rows)
        ev1.record()
        words_ = res[0] if is_find else res
        if use_dist:  # async on RCCL's stream, ordered after the kernel: the next step's kernel overlaps with it
            pending.append(gather_bitmap_async(words_, total_rows, world, rank))
        return res, (ev0, ev1)

    pending = []

    def fence():
        for h in pending:  # every step's bitmap has landed on every rank before the clock stops
            h.wait()
        del pending[:]
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res, _ev = step()
    fence()
    t0 = time.perf_counter()
    events = []
    for _ in range(args.steps):
        res, ev = step()
        events.append(ev)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = sum(a.elapsed_time(b) for a, b in events) / len(events)

    # algorithmic bytes: L*w input bytes + result bytes (1 bit/row; find adds 2 x int32/row)
    per_row = 256 * cw + (8 if is_find else 0)
    bytes_gpu = n_rows * per_row + ((n_rows + 63) // 64) * 8
    bytes_job = total_rows * per_row + ((total_rows + 63) // 64) * 8
    ms_per_step = elapsed / args.steps * 1e3
    matched = None
    if rank == 0:
        from needle_amd.pattern import unpack_bitmap
        w0 = res[0] if is_find else res
        matched = int(unpack_bitmap(w0, n_rows).sum())
    out = {
        "metric": "GB/s haystack scanned (10M x 256-char batch per GPU, DFA table walk)",
        "value": bytes_job / (elapsed / args.steps) / 1e9,
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u8" if cw == 1 else "u16",
        "data": "synthetic",
        "rows_per_s": total_rows / (elapsed / args.steps),
        "matches_per_s": None if matched is None else matched * world / (elapsed / args.steps),
        "config": {"workload": "%s: %s over %d x 256 %s rows per GPU" % (args.workload, what, n_rows, "UTF-16" if cw == 2 else "ASCII"),
                   "rows_total": total_rows, "row_chars": 256, "char_bytes": cw, "parallelism": "row-shard x%d" % world,
                   "result": "bitmap" + ("+start/end int32" if is_find else ""), "pattern": pattern.info()},
        "roofline": {"bound": "hbm", "achieved": bytes_gpu / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": bytes_gpu / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "needle::scan_kernel", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_gpu},
    }
    # HBM traffic per launch comes from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected per
    # MI355X_MICROARCH.md) committed under profiles/ for this exact workload and size; null if none was taken.
    if args.rows == 10_000_000:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s.json" % args.workload)), reverse=True):
            try:
                prof = json.load(open(f))
                out["roofline"]["traffic"] = prof["traffic_bytes_per_launch"]
                out["roofline"]["traffic_source"] = os.path.relpath(f, ROOT)
                break
            except (OSError, KeyError, ValueError):
                pass
    props = torch.cuda.get_device_properties(dev)
    clk_hz = ENGINE_CLOCK_MHZ * 1e6
    out["roofline"]["chars_per_clk_per_cu"] = n_rows * 256 / (kernel_ms * 1e-3) / clk_hz / props.multi_processor_count
    out["roofline"]["device"] = {"name": props.name, "cus": props.multi_processor_count, "clock_mhz_nominal": ENGINE_CLOCK_MHZ}
    if rank == 0 and world == 1 and not args.no_extras:
        ceil = measured_read_ceiling(rows)
        if ceil:
            out["roofline"]["measured_read_ceiling"] = ceil
            out["roofline"]["frac_of_measured_ceiling"] = out["roofline"]["achieved"] / ceil
        mr, exact = must_read_bytes(args.workload if args.regex is None and args.op is None else "custom", pattern, rows, cw)
        out["must_read"] = {"bytes_per_step": mr, "GB/s": mr / (elapsed / args.steps) / 1e9, "exact": exact,
                            "note": "chars the reference loop touches before it stops x bytes/char (SURVEY.md s8d secondary denominator)"}
    if use_dist:
        # the gather reported separately (blocking, nothing overlapped with it) beside the step time in
        # which it IS overlapped with the next step's scan
        from needle_amd.sharding import gather_bitmap
        w0 = res[0] if is_find else res
        gather_bitmap(w0, total_rows, world, rank)
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        for _ in range(10):
            gather_bitmap(w0, total_rows, world, rank)
        torch.cuda.synchronize()
        out["gather"] = {"collective": "all_gather_into_tensor (RCCL)", "bytes_per_rank": int(w0.numel() * 8),
                         "ms_blocking": (time.perf_counter() - g0) / 10 * 1e3,
                         "note": "inside the timed steps it is issued asynchronously and overlaps with the next scan"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload, pattern, rows, op_name)
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
