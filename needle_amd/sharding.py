"""Row sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL on
ROCm, "gloo" in the CPU tests).  Rows are independent units (a Matcher is per haystack, DFAClassBuilder.java:
669-699), so the only communication is the gather of results to rank 0: a fan-in of direct sends (each peer's
own xGMI link to the root), not a ring."""
import torch
import torch.distributed as dist


def shard_range(total_rows, world, rank):
    """Contiguous row block of `rank`; block boundaries are multiples of 64 rows so that every shard's bitmap is
    whole uint64 words.  -> (row0, n_rows)"""
    per = -(-total_rows // world)
    per = -(-per // 64) * 64
    row0 = min(rank * per, total_rows)
    return row0, min(per, total_rows - row0)


def shard_words(total_rows, world, rank):
    row0, n = shard_range(total_rows, world, rank)
    return row0 // 64, (n + 63) // 64


def gather_bitmap(words, total_rows, world, rank, dst=0):
    """Gather the per-shard bitmap words to `dst`.  Returns the full bitmap (int64 words) on dst, None elsewhere."""
    if world == 1:
        return words
    per = -(-(-(-total_rows // world)) // 64)  # words per full shard
    buf = words
    if words.numel() != per:  # last (short / empty) shard: pad to the common size
        buf = torch.zeros(per, dtype=words.dtype, device=words.device)
        buf[:words.numel()] = words
    if rank == dst:
        parts = [torch.empty(per, dtype=words.dtype, device=words.device) for _ in range(world)]
        dist.gather(buf, parts, dst=dst)
        return torch.cat(parts)[:(total_rows + 63) // 64]
    dist.gather(buf, None, dst=dst)
    return None


def gather_rows(values, total_rows, world, rank, dst=0):
    """Gather a per-row int32 result (find start / end) to `dst`."""
    if world == 1:
        return values
    per = -(-(-(-total_rows // world)) // 64) * 64
    buf = values
    if values.numel() != per:
        buf = torch.full((per,), -1, dtype=values.dtype, device=values.device)
        buf[:values.numel()] = values
    if rank == dst:
        parts = [torch.empty(per, dtype=values.dtype, device=values.device) for _ in range(world)]
        dist.gather(buf, parts, dst=dst)
        return torch.cat(parts)[:total_rows]
    dist.gather(buf, None, dst=dst)
    return None
