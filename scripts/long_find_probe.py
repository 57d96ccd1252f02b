"""find() / containedIn() of a literal on 1024 x 1 MiB rows of the Sherlock Holmes text (the speculative-stripe path): for a
rocprofv3 --kernel-trace --stats run that shows where find()'s time goes.  python scripts/long_find_probe.py [regex] [op]"""
import gzip, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from needle_amd.pattern import DFACompiler
rx = sys.argv[1] if len(sys.argv) > 1 else "Sherlock"
opn = sys.argv[2] if len(sys.argv) > 2 else "find"
text = np.frombuffer(gzip.open(os.path.join(ROOT, "tests", "golden", "sherlockholmes.txt.gz")).read(), dtype=np.uint8)
N, ROW = 1024, 1 << 20
rows = torch.from_numpy(np.tile(text, (N * ROW + len(text) - 1) // len(text))[:N * ROW].reshape(N, ROW)).cuda()
p = DFACompiler.compile(rx, "t", 0)
op = p.find_batch if opn == "find" else p.contained_in_batch
for _ in range(3):
    r = op(rows)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    r = op(rows)
e1.record()
torch.cuda.synchronize()
print(rx, opn, "%.3f ms" % (e0.elapsed_time(e1) / 10))
