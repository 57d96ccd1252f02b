#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_longfind; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/find -o t -- python scripts/long_find_probe.py Sherlock find > $O/find.log 2> $O/find.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cont -o t -- python scripts/long_find_probe.py Sherlock contained_in > $O/cont.log 2> $O/cont.err
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*_agent_info.csv" -delete
cat $O/find.log $O/cont.log | grep -v amdgpu
for d in find cont; do echo "== $d"; head -12 $O/$d/t_kernel_stats.csv | cut -c1-200; done
