#!/bin/bash
# one gpurun call: full-size property tests, bench lines with extras, C3 shape sweep
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r1
timeout 1500 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/r1/fullsize.log 2>&1; echo "fullsize rc=$?" 
tail -5 gpurun_out/r1/fullsize.log
for w in c2 c3 c5; do timeout 300 python bench.py --workload $w > gpurun_out/r1/bench_$w.json 2> gpurun_out/r1/bench_$w.err; tail -c 1500 gpurun_out/r1/bench_$w.json; echo; done
for sh in 16x64 12x64 8x64 4x64; do echo "shape $sh"; NEEDLE_SHAPE=$sh timeout 300 python bench.py --workload c3 --no-cpu-baseline --no-extras 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['achieved'])"; done
