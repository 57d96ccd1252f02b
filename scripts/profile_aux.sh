#!/bin/bash
# rocprofv3 --kernel-trace --stats of the kernels DESIGN.md quotes beside the tiled scan: the register-resident
# short-row kernel (16-byte rows) and the stripe path (1000 x 1 MiB rows).  -> gpurun_out/prof_aux/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_aux; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/short16 -o t -- python scripts/short_rows_rate.py 16 > $O/short16.log 2> $O/short16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/short64 -o t -- python scripts/short_rows_rate.py 64 > $O/short64.log 2> $O/short64.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/long -o t -- python scripts/long_rows_rate.py 1000 1 > $O/long.log 2> $O/long.err
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*_agent_info.csv" -delete
cat $O/short16.log $O/short64.log $O/long.log | grep -v amdgpu.ids
for d in short16 short64 long; do echo "== $d"; head -6 $O/$d/t_kernel_stats.csv | cut -c1-170; done
