"""Where do the ~40 us per step go when every step also issues an async RCCL gather?  (run under torchrun, 1 rank)"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, ".")
import bench
from needle_amd.sharding import gather_bitmap_async
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
p, _, words = bench.make_pattern("c2")
n = 10_000_000
rows = bench.make_rows("c2", words, 0, n, "cuda")
use_side = os.environ.get("SIDE_STREAM") == "1"
side = torch.cuda.Stream()
def run(with_gather, steps=30):
    pend = []
    torch.cuda.synchronize()
    t_op = t_g = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        a = time.perf_counter()
        if use_side:
            with torch.cuda.stream(side):
                w = p.contained_in_batch(rows)
        else:
            w = p.contained_in_batch(rows)
        b = time.perf_counter()
        if with_gather:
            if use_side:
                with torch.cuda.stream(side):
                    pend.append(gather_bitmap_async(w, n, 1, 0))
            else:
                pend.append(gather_bitmap_async(w, n, 1, 0))
        c = time.perf_counter()
        t_op += b - a; t_g += c - b
    for h in pend: h.wait()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("gather=%s side=%s: %.1f us/step wall, cpu in op %.1f us, cpu in gather %.1f us" % (with_gather, use_side, el / steps * 1e6, t_op / steps * 1e6, t_g / steps * 1e6))
for _ in range(2):
    run(False); run(True)
dist.destroy_process_group()
