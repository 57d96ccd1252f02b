import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/stream_pattern.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "stream_pattern.hip")])
L = ctypes.CDLL(so)
L.launch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
n = 2_560_000_000
buf = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
out = torch.zeros(4, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for mode in (0, 1):
    for blocks, threads in ((256, 1024), (512, 512), (256, 512), (1024, 256), (2048, 256)):
        for _ in range(3):
            L.launch(buf.data_ptr(), n, out.data_ptr(), blocks, threads, mode, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.launch(buf.data_ptr(), n, out.data_ptr(), blocks, threads, mode, s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("mode", mode, "blocks", blocks, "threads", threads, "%.3f ms  %.0f GB/s" % (ms, n / ms / 1e6))
