"""Is the per-step cost of the async gather RCCL's, or that of ANY cross-stream dependency per step?"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, ".")
import bench
from needle_amd.sharding import gather_bitmap_async
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
p, _, words = bench.make_pattern("c2")
n = 10_000_000
rows = bench.make_rows("c2", words, 0, n, "cuda")
side = torch.cuda.Stream()
out = torch.empty(n // 64 + 1, dtype=torch.int64, device="cuda")
def run(mode, steps=40):
    pend = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prev = None
    for _ in range(steps):
        w = p.contained_in_batch(rows)
        if mode == "rccl":
            pend.append(gather_bitmap_async(w, n, 1, 0))
        elif mode == "rccl_delayed":
            if prev is not None: pend.append(gather_bitmap_async(prev, n, 1, 0))
            prev = w
        elif mode == "side_copy":
            ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev); out[:w.numel()].copy_(w)
            w.record_stream(side)
        elif mode == "event_only":
            ev = torch.cuda.Event(); ev.record()
    for h in pend: h.wait()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("%-14s %.1f us/step" % (mode, el / steps * 1e6))
for _ in range(2):
    for m in ("none", "event_only", "side_copy", "rccl", "rccl_delayed"): run(m)
dist.destroy_process_group()
