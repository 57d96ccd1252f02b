"""Row sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL on
ROCm, "gloo" in the CPU tests).  Rows are independent units (a Matcher is per haystack, DFAClassBuilder.java:
669-699), so there is no data-path collective.  What moves, per step, are results only:

  * the per-shard result BITMAP (156 KB per rank at 10M rows over 8 GPUs): one all-gather, every rank ends up with
    the whole bitmap;
  * for find(), the per-row start / end: a FAN-IN to rank 0 -- every peer sends over its own direct xGMI link to the
    root (7 links in parallel).  An all-gather would push the same bytes through every link of a ring and serve no
    one: only the host side of rank 0 consumes them.  Rows of up to 65 534 chars travel as ONE dword per row (two
    16-bit halves, 0xFFFF = no match: 5 MB per rank at 10M rows over 8 GPUs instead of 10) and are unpacked on rank 0
    when somebody asks for them.

Both are issued asynchronously (RCCL's own stream, ordered after the scan kernel), so the next step's scan overlaps
with them.  `ShardedScan` is the one implementation of a step: bench.py drives it with the HIP kernels, the gloo
tests with the CPU oracle."""
import torch
import torch.distributed as dist


PACK16_MAX_ROW_LEN = 65534  # needle_pack_start_end16_dev: offsets 0 .. 65534 fit a half word, 0xFFFF is "no match"


def shard_range(total_rows, world, rank):
    """Contiguous row block of `rank`; block boundaries are multiples of 64 rows so that every shard's bitmap is
    whole uint64 words.  -> (row0, n_rows)"""
    per = -(-total_rows // world)
    per = -(-per // 64) * 64
    row0 = min(rank * per, total_rows)
    return row0, min(per, total_rows - row0)


def _words_per_shard(total_rows, world):
    return -(-(-(-total_rows // world)) // 64)


def _dist_on():
    return dist.is_available() and dist.is_initialized()


class _Pending:
    """Handle of an in-flight gather: .wait() -> the gathered tensor (or None on the ranks that do not receive)."""

    def __init__(self, work, out, n, keep):
        self.work, self.out, self.n, self.keep = work, out, n, keep

    def wait(self):
        if self.work is not None:
            self.work.wait()
        out = self.out
        if out is None:
            return None
        if isinstance(out, list):
            out = torch.cat(out)
        return out[:self.n]


class _EventPending:
    """The same for a gather issued on a side stream: wait() orders the current stream behind it."""

    def __init__(self, event, out, n):
        self.event, self.out, self.n = event, out, n

    def wait(self):
        if self.event is not None:  # (None: the gather was issued on the current stream itself)
            torch.cuda.current_stream().wait_event(self.event)
        return None if self.out is None else self.out[:self.n]


def gather_bitmap_async(words, total_rows, world, rank, out=None):
    """Start the all-gather of the per-shard bitmap words; returns a handle whose wait() yields the full bitmap on
    every rank.  The caller may launch further work on its stream before waiting.  `words` may already be the padded
    per-shard buffer (ShardedScan's are)."""
    n_words = (total_rows + 63) // 64
    if world == 1 and not _dist_on():
        return _Pending(None, words, n_words, None)
    per = _words_per_shard(total_rows, world)
    buf = words
    if words.numel() != per:  # last (short / empty) shard: pad to the common size
        buf = torch.zeros(per, dtype=words.dtype, device=words.device)
        buf[:words.numel()] = words
    if buf.is_cuda:
        if out is None:
            out = torch.empty(per * world, dtype=words.dtype, device=words.device)
        work = dist.all_gather_into_tensor(out, buf, async_op=True)
    else:  # gloo (CPU tests)
        out = [torch.empty(per, dtype=words.dtype) for _ in range(world)]
        work = dist.all_gather(out, buf, async_op=True)
    return _Pending(work, out, n_words, buf)


def gather_bitmap(words, total_rows, world, rank):
    """Blocking form: the full bitmap on every rank."""
    return gather_bitmap_async(words, total_rows, world, rank).wait()


def gather_rows_to_root_async(values, total_rows, world, rank, fill=-1, out=None):
    """Fan-in of a per-row int32 result (find start / end) to rank 0: wait() -> the full array on rank 0, None on the
    other ranks.  `values` may already be the padded per-shard buffer."""
    if world == 1 and not _dist_on():
        return _Pending(None, values, total_rows, None)
    per = _words_per_shard(total_rows, world) * 64
    buf = values
    if values.numel() != per:
        buf = torch.full((per,), fill, dtype=values.dtype, device=values.device)
        buf[:values.numel()] = values
    parts = None
    if rank == 0:
        if out is None:
            out = torch.empty(per * world, dtype=values.dtype, device=values.device)
        parts = list(out.view(world, per).unbind(0))
    work = dist.gather(buf, gather_list=parts, dst=0, async_op=True)
    return _Pending(work, out if rank == 0 else None, total_rows, (buf, parts))


def gather_rows(values, total_rows, world, rank, fill=-1):
    """All-gather a per-row int32 result; returns the full array on every rank (tests / small batches: the step
    itself uses the fan-in above)."""
    if world == 1 and not _dist_on():
        return values
    per = _words_per_shard(total_rows, world) * 64
    buf = values
    if values.numel() != per:
        buf = torch.full((per,), fill, dtype=values.dtype, device=values.device)
        buf[:values.numel()] = values
    if buf.is_cuda:
        out = torch.empty(per * world, dtype=values.dtype, device=values.device)
        dist.all_gather_into_tensor(out, buf)
    else:
        parts = [torch.empty(per, dtype=values.dtype) for _ in range(world)]
        dist.all_gather(parts, buf)
        out = torch.cat(parts)
    return out[:total_rows]


class ShardedScan:
    """One step of the row-sharded job (config C4): scan this rank's rows, then bring the results to rank 0.

    scan(bitmap, start, end): writes this shard's verdicts into the given buffers (int64 bitmap words; int32 start /
    end, both None unless `is_find`); the buffers are the padded per-shard send buffers themselves, so a step moves
    no byte more than the collectives do.  matches / containedIn: ONE all-gather of the bitmap words (every rank ends
    up with the whole bitmap).  find: start, end and the bitmap words sit back to back in ONE send buffer per rank and
    go to rank 0 in ONE fan-in gather -- one collective call per step either way (on a 1.25M-row shard the scan takes
    ~0.1 ms; three separately issued collectives cost the host more than that).  `n_buffers` result sets rotate so that
    step k + 1 may scan while the gather of step k is still in flight."""

    def __init__(self, scan, total_rows, world, rank, is_find, device, n_buffers=2, comm=None, overlap=None, pack16=False,
                 max_row_len=None, scan_packed=None):
        """comm: a needle_amd.multi.RankComm -- the gather is then ONE call into the library's own RCCL communicator (GPU
        runs); without it the gather goes through torch.distributed (any backend: the gloo tests).
        overlap (with comm): issue the gather on a side stream so that the next scan runs beside it.  That costs two
        cross-stream event handshakes per step -- ~25 us of queue latency, measured on 1.25M-row shards: c2 88 us per step
        against 63 us with the gather simply queued behind the kernel on the scan's own stream (kernel 55 us), find 120
        against 101 us -- so the default is the scan's own stream."""
        self.scan, self.total_rows, self.world, self.rank, self.is_find = scan, total_rows, world, rank, is_find
        self.comm, self.side = comm, None
        # pack16 (find): start / end cross the links as one dword per row, two 16-bit halves with 0xFFFF = no match -- exact
        # only for rows of at most 65 534 chars (an end of 65 535 would read as "no match", longer offsets would be cut).
        # The caller states the longest row (max_row_len); longer rows, or no statement at all, keep the 8-byte form.
        self.pack16 = bool(pack16) and is_find and max_row_len is not None and int(max_row_len) <= PACK16_MAX_ROW_LEN
        # scan_packed(bitmap, packed) (with pack16): the scan stores the dword form itself (needle_find_packed16_dev) straight into
        # the send buffer -- no int32 start / end arrays, no pack pass between the scan and the gather
        self.scan_packed = scan_packed if self.pack16 else None
        self.overlap = bool(overlap)
        if comm is not None and self.overlap:
            self.side = torch.cuda.Stream(device=device)
        self.row0, self.n_rows = shard_range(total_rows, world, rank)
        self.dist = _dist_on() or comm is not None
        self.per_words = _words_per_shard(total_rows, world) if world > 1 or self.dist else (total_rows + 63) // 64
        self.per_rows = per_rows = self.per_words * 64
        self.sets, self.k = [], 0
        for _ in range(n_buffers):
            s = {"pending": None}
            if is_find and self.pack16:
                # send buffer [start | end as 16-bit halves: per_rows int32 | bitmap: per_words int64 viewed as int32 pairs];
                # the scan writes int32 start / end next to it, the pack kernel (or three torch ops on the CPU) fills it
                buf = torch.full((per_rows + 2 * self.per_words,), -1, dtype=torch.int32, device=device)
                buf[per_rows:] = 0
                s["buf"] = buf
                if self.scan_packed is None:
                    s["start"] = torch.full((per_rows,), -1, dtype=torch.int32, device=device)
                    s["end"] = torch.full((per_rows,), -1, dtype=torch.int32, device=device)
                s["packed"] = buf[:per_rows]
                s["bitmap"] = buf[per_rows:].view(torch.int64)
                if (self.dist or comm is not None) and rank == 0:
                    s["all"] = torch.empty((world, buf.numel()), dtype=torch.int32, device=device)
            elif is_find:
                # [start: per_rows int32 | end: per_rows int32 | bitmap: per_words int64 viewed as int32 pairs]
                buf = torch.full((2 * per_rows + 2 * self.per_words,), -1, dtype=torch.int32, device=device)
                buf[2 * per_rows:] = 0
                s["buf"] = buf
                s["start"], s["end"] = buf[:per_rows], buf[per_rows:2 * per_rows]
                s["bitmap"] = buf[2 * per_rows:].view(torch.int64)
                if (self.dist or comm is not None) and rank == 0:  # the receive buffer is part of the set: no allocation inside a step
                    s["all"] = torch.empty((world, buf.numel()), dtype=torch.int32, device=device)
            else:
                s["bitmap"] = torch.zeros(self.per_words, dtype=torch.int64, device=device)
                if (self.dist or comm is not None) and torch.device(device).type == "cuda":
                    s["bitmap_all"] = torch.empty(self.per_words * world, dtype=torch.int64, device=device)
            self.sets.append(s)

    def scan_only(self):
        """The scan of this rank's shard into the current buffer set (no communication)."""
        s = self.sets[self.k % len(self.sets)]
        if self.n_rows:
            if self.scan_packed is not None:
                self.scan_packed(s["bitmap"], s["packed"])
            else:
                self.scan(s["bitmap"], s.get("start"), s.get("end"))
        return s

    def step(self, events=None):
        """Scan + start the gather; returns the buffer set, whose "pending" handle wait() completes.  events: an
        optional pair of stream events recorded right before and right after the scan (the gather runs on the
        collective library's own stream and is not between them)."""
        s = self.sets[self.k % len(self.sets)]
        if s["pending"] is not None:  # this set's previous gather must have left before the scan overwrites it
            s["pending"].wait()
        if events is not None:
            events[0].record()
        self.scan_only()
        if events is not None:
            events[1].record()
        self.k += 1
        if self.pack16 and self.dist and self.scan_packed is None:
            self._pack(s)
        if not self.dist:
            s["pending"] = _Pending(None, None, 0, None)
        elif self.comm is not None:
            # the library's communicator: right behind the scan on its stream, or on a side stream ordered after it
            cur = torch.cuda.current_stream()
            if self.side is not None:
                self.side.wait_stream(cur)
            st = (self.side if self.side is not None else cur).cuda_stream
            if self.is_find:
                self.comm.gather_i32(s["buf"], s.get("all"), st)
            else:
                self.comm.all_gather_u64(s["bitmap"], s["bitmap_all"], st)
            ev = None
            if self.side is not None:
                ev = torch.cuda.Event()
                ev.record(self.side)
            s["pending"] = _EventPending(ev, None if self.is_find else s["bitmap_all"], (self.total_rows + 63) // 64)
        elif self.is_find:
            parts = list(s["all"].unbind(0)) if self.rank == 0 else None
            s["pending"] = _Pending(dist.gather(s["buf"], gather_list=parts, dst=0, async_op=True), None, 0, parts)
        else:
            s["pending"] = gather_bitmap_async(s["bitmap"], self.total_rows, self.world, self.rank, out=s.get("bitmap_all"))
        return s

    def _pack(self, s):
        if s["packed"].is_cuda:
            from . import _lib
            from .pattern import _check
            _check(_lib.lib().needle_pack_start_end16_dev(s["start"].data_ptr(), s["end"].data_ptr(), self.per_rows, s["packed"].data_ptr(),
                                                          torch.cuda.current_stream().cuda_stream))
        else:  # the gloo tests
            torch.bitwise_or(s["start"] & 0xFFFF, s["end"] << 16, out=s["packed"])

    @staticmethod
    def _unpack(packed):
        if packed.is_cuda:
            from . import _lib
            from .pattern import _check
            packed = packed.contiguous()
            start, end = torch.empty_like(packed), torch.empty_like(packed)
            _check(_lib.lib().needle_unpack_start_end16_dev(packed.data_ptr(), packed.numel(), start.data_ptr(), end.data_ptr(),
                                                            torch.cuda.current_stream().cuda_stream))
            return start, end
        lo, hi = packed & 0xFFFF, (packed >> 16) & 0xFFFF
        minus1 = torch.full_like(lo, -1)
        return torch.where(lo == 0xFFFF, minus1, lo), torch.where(hi == 0xFFFF, minus1, hi)

    def wait(self, s):
        """-> (full bitmap words, start, end).  matches / containedIn: the bitmap on every rank.  find: all three on
        rank 0 (views into the gathered buffer, rows in shard order), None elsewhere."""
        h, s["pending"] = s["pending"], None
        n_words = (self.total_rows + 63) // 64
        if not self.dist:
            if h is not None:
                h.wait()
            if self.scan_packed is not None:  # (one rank, no gather: the packed scan's dwords are unpacked here)
                start, end = self._unpack(s["packed"][:self.total_rows])
                return s["bitmap"][:n_words], start, end
            return s["bitmap"][:n_words], (s["start"][:self.total_rows] if self.is_find else None), (s["end"][:self.total_rows] if self.is_find else None)
        if not self.is_find:
            return h.wait(), None, None
        h.wait()
        if self.rank != 0:
            return None, None, None
        g, pr = s["all"], self.per_rows
        if self.pack16:
            start, end = self._unpack(g[:, :pr].reshape(-1)[:self.total_rows])
            bitmap = g[:, pr:].contiguous().view(torch.int64).reshape(-1)[:n_words]
            return bitmap, start, end
        start = g[:, :pr].reshape(-1)[:self.total_rows]
        end = g[:, pr:2 * pr].reshape(-1)[:self.total_rows]
        bitmap = g[:, 2 * pr:].contiguous().view(torch.int64).reshape(-1)[:n_words]
        return bitmap, start, end

    def drain(self):
        for s in self.sets:
            if s["pending"] is not None:
                s["pending"].wait()
                s["pending"] = None
