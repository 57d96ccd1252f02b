#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r1/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -5 gpurun_out/r1/gpu_tests.log
for w in c2 c3 c5; do timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['roofline']['kernel_ms'], d['roofline']['achieved'])"; done
for sh in 16x64 12x64 12x128 8x128; do echo "c3 shape $sh"; NEEDLE_SHAPE=$sh timeout 300 python bench.py --workload c3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['achieved'])"; done
