import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
w = torch.zeros(156250 // 8, dtype=torch.int64, device="cuda")
out = torch.empty_like(w)
x = torch.zeros(64 << 20, dtype=torch.float32, device="cuda")
def busy():  # ~0.5 ms of GPU work on the current stream
    x.mul_(1.0001)
for name, fn in (("none", lambda: None),
                 ("all_gather_into_tensor async", lambda: dist.all_gather_into_tensor(out, w, async_op=True)),
                 ("all_gather_into_tensor sync", lambda: dist.all_gather_into_tensor(out, w))):
    for _ in range(5): busy(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hs = []
    for _ in range(50):
        busy(); hs.append(fn())
    th = time.perf_counter() - t0
    for h in hs:
        if h is not None: h.wait()
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    print("%-32s host %.1f us/iter   total %.1f us/iter" % (name, th / 50 * 1e6, t1 / 50 * 1e6))
dist.destroy_process_group()
