"""N > 1 path on CPU: world_size-2 (and 3) gloo process groups exercising needle_amd/sharding.py -- contiguous
row blocks on 64-row boundaries, the bitmap / per-row gathers to rank 0 -- against the unsharded CPU oracle.
(The per-shard verdicts come from the oracle here; on the GPU box the same code path carries kernel output.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_snapshot


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pack(bits):
    pad = (-len(bits)) % 64
    b = np.concatenate([bits.astype(np.uint8), np.zeros(pad, dtype=np.uint8)])
    return torch.from_numpy(np.packbits(b, bitorder="little").view(np.int64).copy())


def _worker(rank, world, port, total_rows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from needle_amd import workload as W
        from needle_amd.sharding import ShardedScan, gather_bitmap, gather_rows, shard_range
        from oracle.walker import OraclePattern
        o = OraclePattern.from_fixture(load_snapshot("DigitPlus"), backwards_as_dfa=True)
        row0, n = shard_range(total_rows, world, rank)
        assert row0 % 64 == 0 or n == 0
        rows = W.digits_batch(np, row0, n, 64) if n else np.zeros((0, 64), dtype=np.uint8)

        # ShardedScan is the step bench.py times (there the scan is the HIP kernel, here the CPU oracle): scan into the
        # padded per-shard buffers, all-gather of the bitmap, fan-in of start / end to rank 0 -- two steps in flight
        def scan(bitmap, start, end):
            m, s, e = o.batch_find(rows)
            bitmap[:(n + 63) // 64] = _pack(m)
            start[:n] = torch.from_numpy(s.astype(np.int32))
            end[:n] = torch.from_numpy(e.astype(np.int32))

        sh = ShardedScan(scan, total_rows, world, rank, True, "cpu", n_buffers=2)
        assert (sh.row0, sh.n_rows) == (row0, n)
        s1 = sh.step()
        s2 = sh.step()
        full, st, en = sh.wait(s1)
        full2, st2, en2 = sh.wait(s2)
        if rank == 0:  # find: start | end | bitmap fan in to rank 0 in one gather
            assert full.shape[0] == (total_rows + 63) // 64 and (full == full2).all()
            assert st.shape[0] == total_rows and (st == st2).all() and (en == en2).all()
        else:
            assert full is None and st is None and en is None
        # the same step with start / end travelling as one dword per row (two 16-bit halves, 0xFFFF = -1)
        shp = ShardedScan(scan, total_rows, world, rank, True, "cpu", n_buffers=2, pack16=True, max_row_len=rows.shape[1] if n else 256)
        assert shp.pack16
        sp = shp.step()
        assert sp["buf"].numel() == shp.per_rows + 2 * shp.per_words  # half the start / end bytes on the wire
        # rows the halves cannot hold (or no stated length at all) keep the 8-byte form instead of truncating silently
        assert not ShardedScan(scan, total_rows, world, rank, True, "cpu", pack16=True, max_row_len=65535).pack16
        assert not ShardedScan(scan, total_rows, world, rank, True, "cpu", pack16=True).pack16
        fullp, stp, enp = shp.wait(sp)
        if rank == 0:
            assert (fullp == full).all() and (stp == st).all() and (enp == en).all()
        else:
            assert fullp is None and stp is None and enp is None
        # ... and with the scan storing the dword form itself, straight into the send buffer (needle_find_packed16_dev on the GPU;
        # here the oracle's results packed the same way): no int32 arrays, no pack pass, the same gathered results
        def scan_packed(bitmap, packed):
            m, s, e = o.batch_find(rows)
            bitmap[:(n + 63) // 64] = _pack(m)
            packed[:n] = torch.from_numpy(((s.astype(np.int64) & 0xFFFF) | ((e.astype(np.int64) & 0xFFFF) << 16)).astype(np.uint32).view(np.int32))

        shd = ShardedScan(scan, total_rows, world, rank, True, "cpu", n_buffers=2, pack16=True, max_row_len=rows.shape[1] if n else 256,
                          scan_packed=scan_packed)
        assert shd.pack16 and "start" not in shd.sets[0]
        fulld, std, end_ = shd.wait(shd.step())
        if rank == 0:
            assert (fulld == full).all() and (std == st).all() and (end_ == en).all()
        else:
            assert fulld is None and std is None and end_ is None
        assert ShardedScan(scan, total_rows, world, rank, True, "cpu", pack16=True, max_row_len=70000, scan_packed=scan_packed).scan_packed is None
        # the contained_in step (bitmap only) and the plain helpers
        bits = o.batch_contained_in(rows) if n else np.zeros(0, dtype=bool)

        def scan_c(bitmap, start, end):
            assert start is None and end is None
            bitmap[:(n + 63) // 64] = _pack(bits)

        shc = ShardedScan(scan_c, total_rows, world, rank, False, "cpu")
        full_c, _, _ = shc.wait(shc.step())
        assert (full_c == gather_bitmap(_pack(bits), total_rows, world, rank)).all()
        _, _, e = o.batch_find(rows) if n else (None, None, np.zeros(0, np.int32))
        # bench.py's self-check of an N > 1 run (popcount of the gathered bitmap, position-weighted checksum of the gathered
        # start / end against what the ranks computed): passes on the real gather, trips on a gather that delivers the
        # shards in the wrong order
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import verify_gather
        for find_op, step_obj in ((True, sh), (True, shp), (True, shd), (False, shc)):
            chk = verify_gather(step_obj, "cpu", row0, n, total_rows, find_op, rank)
            assert chk["ok"] and chk["popcount_gathered" if (rank == 0 or not find_op) else "popcount_ranks"] == chk["popcount_ranks"]
        if world > 1 and total_rows > 64 * world:
            real_wait = sh.wait

            def swapped(s_):  # rank 0 sees the shards' rows rotated by one shard
                f, a, b = real_wait(s_)
                if a is None:
                    return f, a, b
                k = sh.per_rows
                return f, torch.roll(a, k), torch.roll(b, k)
            sh.wait = swapped
            assert not verify_gather(sh, "cpu", row0, n, total_rows, True, rank)["ok"]
            sh.wait = real_wait
        ends_all = gather_rows(torch.from_numpy(e.astype(np.int32)), total_rows, world, rank)
        if rank == 0:
            assert (ends_all == en).all()
            q.put((full.numpy().copy(), full_c.numpy().copy(), st.numpy().copy(), en.numpy().copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total_rows", [(2, 1000), (3, 64 * 7 + 5), (2, 1)])
def test_row_sharding_and_gather(world, total_rows, oracle_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, full_c, starts, ends = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    from needle_amd import workload as W
    from needle_amd.pattern import unpack_bitmap
    from oracle.walker import OraclePattern
    o = OraclePattern.from_fixture(load_snapshot("DigitPlus"), backwards_as_dfa=True)
    rows = W.digits_batch(np, 0, total_rows, 64)
    want = o.batch_contained_in(rows)
    assert full_c.shape[0] == (total_rows + 63) // 64
    assert (unpack_bitmap(full_c, total_rows) == want).all()
    m, s, e = o.batch_find(rows)
    assert (unpack_bitmap(full, total_rows) == m).all()
    assert (ends == e).all() and (starts == s).all()


def test_shard_ranges_partition_the_batch():
    from needle_amd.sharding import shard_range
    for total in (0, 1, 63, 64, 65, 1000, 10_000_000, 80_000_000):
        for world in (1, 2, 4, 8):
            nxt = 0
            for r in range(world):
                row0, n = shard_range(total, world, r)
                assert row0 == min(nxt, total) and n >= 0
                assert row0 % 64 == 0 or n == 0
                nxt = row0 + n
            assert nxt == total
