#!/bin/bash
cd "$GRAFT_REPO_ROOT"
q() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'issue us', round(d['host_issue_us_per_step'],1), d.get('scan_ms'), d.get('gather_ms'))"; }
T="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --also none --no-cpu-baseline --no-extras"
for w in c2 c3; do for o in on off; do
  timeout 300 $T --workload $w --rows 1250000 --steps 300 --warmup 30 --overlap $o 2>/dev/null | q "$w 1.25M overlap=$o"
  timeout 300 $T --workload $w --steps 30 --overlap $o 2>/dev/null | q "$w 10M overlap=$o"
done; done
python -m pytest tests/test_gpu_multi_device.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
