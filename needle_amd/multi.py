"""Row sharding over several devices from ONE host process: the ctypes mirror of needle_multi_* (include/needle_hip.h).
This is the entry a JVM host uses (bindings/java GpuPattern.*Batch with several devices); the one-process-per-GPU
form used by bench.py lives in needle_amd/sharding.py."""
import ctypes

import numpy as np

from . import _lib
from ._lib import BatchView
from .pattern import _check

OPS = {"matches": 0, "contained_in": 1, "find": 2}


class MultiDevice:
    def __init__(self, devices, loopback=False):
        self.devices = [int(d) for d in devices]
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        h = ctypes.c_void_p()
        _check(_lib.lib().needle_multi_create(arr, len(self.devices), 1 if loopback else 0, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and _lib._lib is not None:
            _lib._lib.needle_multi_destroy(h)

    def transport(self):
        """What carries the gather: "rccl", "peer-copy: <why RCCL is not used>" or "local" (every shard on one device)."""
        return _lib.lib().needle_multi_transport(self._h).decode()

    def scan(self, pattern, op, shards, lengths=None):
        """shards: one 2-D device tensor per device (every one but the last with a multiple of 64 rows), lengths: None or
        one int32 device tensor per shard.  -> (bitmap int64 words, start, end) on the root device (start / end None
        unless op == "find"), complete when this returns."""
        import torch
        n = len(self.devices)
        assert len(shards) == n
        views = (BatchView * n)()
        total = 0
        for g, t in enumerate(shards):
            assert t.is_cuda and t.dim() == 2 and t.is_contiguous() and t.device.index == self.devices[g]
            v = views[g]
            v.rows, v.char_width, v.n_rows, v.row_stride, v.row_len = t.data_ptr(), t.element_size(), t.shape[0], t.shape[1], t.shape[1]
            if lengths is not None and lengths[g] is not None:
                assert lengths[g].dtype == torch.int32 and lengths[g].shape == (t.shape[0],) and lengths[g].device == t.device
                v.lengths = lengths[g].data_ptr()
            total += t.shape[0]
        root = torch.device("cuda", self.devices[0])
        words = torch.empty((total + 63) // 64, dtype=torch.int64, device=root)
        st = en = None
        if op == "find":
            st = torch.empty(total, dtype=torch.int32, device=root)
            en = torch.empty(total, dtype=torch.int32, device=root)
        # the shards were produced on torch's streams of THEIR devices; the library reads them on its own (non-blocking)
        # streams: every device involved has to be idle first (torch.cuda.synchronize() alone waits for the current one)
        for d in sorted(set(self.devices)):
            torch.cuda.synchronize(d)
        _check(_lib.lib().needle_multi_scan(self._h, pattern._h, OPS[op], views, words.data_ptr(),
                                            st.data_ptr() if st is not None else None, en.data_ptr() if en is not None else None))
        _check(_lib.lib().needle_multi_sync(self._h))
        return words, st, en

    def scan_host(self, pattern, op, rows, lengths=None):
        """A host batch (2-D uint8 / uint16 numpy array) split over the devices -> (bitmap uint64 words, start, end)."""
        rows = np.ascontiguousarray(rows)
        if rows.dtype == np.int16:
            rows = rows.view(np.uint16)
        assert rows.ndim == 2 and rows.dtype in (np.uint8, np.uint16)
        n, stride = rows.shape
        v = BatchView()
        v.rows, v.char_width, v.n_rows, v.row_stride, v.row_len = rows.ctypes.data, rows.dtype.itemsize, n, stride, stride
        if lengths is not None:
            lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
            v.lengths = lengths.ctypes.data
        words = np.zeros((n + 63) // 64, dtype=np.uint64)
        st = np.full(n, -1, dtype=np.int32) if op == "find" else None
        en = np.full(n, -1, dtype=np.int32) if op == "find" else None
        _check(_lib.lib().needle_scan_host_multi(self._h, pattern._h, OPS[op], ctypes.byref(v), words.ctypes.data,
                                                 st.ctypes.data if st is not None else None, en.ctypes.data if en is not None else None))
        return words, st, en


class RankComm:
    """One process per device: this rank's RCCL communicator inside the library (needle_multi_create_rank) and its two
    gathers.  The id rank 0 creates travels over torch.distributed (any backend); after that a step's communication is
    ONE C call on a stream of the caller's choice -- no per-step Python collective machinery."""

    def __init__(self, unique_id, rank, world, device_index):
        h = ctypes.c_void_p()
        buf = (ctypes.c_ubyte * 128).from_buffer_copy(unique_id)
        _check(_lib.lib().needle_multi_create_rank(buf, int(rank), int(world), int(device_index), ctypes.byref(h)))
        self._h, self.rank, self.world = h, rank, world

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and _lib._lib is not None:
            _lib._lib.needle_multi_destroy(h)

    @classmethod
    def from_torch_distributed(cls, device):
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        on_gpu = dist.get_backend() == "nccl"
        idt = torch.zeros(128, dtype=torch.uint8, device=device if on_gpu else "cpu")
        err = None
        if rank == 0:
            try:
                raw = (ctypes.c_ubyte * 128)()
                _check(_lib.lib().needle_multi_unique_id(raw))
                idt.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
            except Exception as e:  # noqa: BLE001 -- the broadcast below must still happen: the other ranks are waiting in it
                err = e
        dist.broadcast(idt, src=0)
        if err is not None:
            raise err
        if not bool(idt.any()):
            raise RuntimeError("rank 0 could not create an RCCL unique id")
        return cls(bytes(idt.cpu().numpy().tobytes()), rank, world, torch.device(device).index)

    def all_gather_u64(self, send, recv, stream):
        _check(_lib.lib().needle_multi_all_gather_u64(self._h, send.data_ptr(), send.numel(), recv.data_ptr(), stream))

    def gather_i32(self, send, recv_root, stream):
        _check(_lib.lib().needle_multi_gather_i32(self._h, send.data_ptr(), send.numel(),
                                                  recv_root.data_ptr() if recv_root is not None else None, stream))
