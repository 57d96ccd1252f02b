"""Latency of the single-haystack Matcher mirror (one 1-row batch per call)."""
import sys, time
sys.path.insert(0, ".")
from needle_amd.pattern import DFACompiler
p = DFACompiler.compile("http://.+")
s = "see http://www.example.com/index.html for details"
m = p.matcher(s); m.find()
t0 = time.perf_counter()
for _ in range(500):
    m = p.matcher(s)
    assert m.find() and (m.start(), m.end()) == (4, 49)
dt = time.perf_counter() - t0
print("matcher(s).find(): %.1f us per call" % (dt / 500 * 1e6))
t0 = time.perf_counter()
for _ in range(500):
    assert p.matcher(s).containedIn()
print("matcher(s).containedIn(): %.1f us per call" % ((time.perf_counter() - t0) / 500 * 1e6))
