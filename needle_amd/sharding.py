"""Row sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL on
ROCm, "gloo" in the CPU tests).  Rows are independent units (a Matcher is per haystack, DFAClassBuilder.java:
669-699), so there is no data-path collective: the only communication is ONE small collective per step that brings
the per-shard result bitmap (156 KB per rank at 10M rows) to every rank over xGMI.  It is issued asynchronously
(RCCL's own stream, ordered after the scan kernel) so that the next step's kernel overlaps with it."""
import torch
import torch.distributed as dist


def shard_range(total_rows, world, rank):
    """Contiguous row block of `rank`; block boundaries are multiples of 64 rows so that every shard's bitmap is
    whole uint64 words.  -> (row0, n_rows)"""
    per = -(-total_rows // world)
    per = -(-per // 64) * 64
    row0 = min(rank * per, total_rows)
    return row0, min(per, total_rows - row0)


def _words_per_shard(total_rows, world):
    return -(-(-(-total_rows // world)) // 64)


class _Pending:
    """Handle of an in-flight gather: .wait() -> full bitmap (int64 words, ceil(total_rows / 64))."""

    def __init__(self, work, out, n_words, keep):
        self.work, self.out, self.n_words, self.keep = work, out, n_words, keep

    def wait(self):
        if self.work is not None:
            self.work.wait()
        out = self.out
        if isinstance(out, list):
            out = torch.cat(out)
        return out[:self.n_words]


def gather_bitmap_async(words, total_rows, world, rank):
    """Start the all-gather of the per-shard bitmap words; returns a handle whose wait() yields the full bitmap on
    every rank.  The caller may launch further work on its stream before waiting."""
    n_words = (total_rows + 63) // 64
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return _Pending(None, words, n_words, None)
    per = _words_per_shard(total_rows, world)
    buf = words
    if words.numel() != per:  # last (short / empty) shard: pad to the common size
        buf = torch.zeros(per, dtype=words.dtype, device=words.device)
        buf[:words.numel()] = words
    if buf.is_cuda:
        out = torch.empty(per * world, dtype=words.dtype, device=words.device)
        work = dist.all_gather_into_tensor(out, buf, async_op=True)
    else:  # gloo (CPU tests)
        out = [torch.empty(per, dtype=words.dtype) for _ in range(world)]
        work = dist.all_gather(out, buf, async_op=True)
    return _Pending(work, out, n_words, buf)


def gather_bitmap(words, total_rows, world, rank):
    """Blocking form: the full bitmap on every rank."""
    return gather_bitmap_async(words, total_rows, world, rank).wait()


def gather_rows(values, total_rows, world, rank, fill=-1):
    """All-gather a per-row int32 result (find start / end); returns the full array on every rank."""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
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
