#!/bin/bash
# How much does the per-step async RCCL gather cost next to the persistent scan kernel?  (world = 1 under torchrun)
cd "$GRAFT_REPO_ROOT"
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['gather']['ms_blocking'],4))"; }
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-dist', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
run "default"
NEEDLE_RESERVE_CUS=4 run "reserve4"
NEEDLE_RESERVE_CUS=8 run "reserve8"
NEEDLE_RESERVE_CUS=4 NCCL_MAX_NCHANNELS=4 run "reserve4+ch4"
NEEDLE_RESERVE_CUS=8 NCCL_MAX_NCHANNELS=4 run "reserve8+ch4"
NEEDLE_RESERVE_CUS=4 NCCL_MAX_NCHANNELS=2 run "reserve4+ch2"
NEEDLE_RESERVE_CUS=16 NCCL_MAX_NCHANNELS=8 run "reserve16+ch8"
NEEDLE_RESERVE_CUS=32 run "reserve32"
