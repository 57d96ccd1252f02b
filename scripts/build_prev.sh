#!/bin/bash
# builds the library as of git revision $1 (default HEAD) into needle_amd/libneedle_hip_prev.so for scripts/ab.sh
REV=${1:-HEAD}
D=$(mktemp -d)
git archive $REV needle_amd/csrc include | tar -x -C $D
(cd $D/needle_amd/csrc && for f in $(ls *.hip *.cpp | grep -v stream_probe); do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-variable -c $f -o $f.o & done; wait
 /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/needle_amd/libneedle_hip_${NEEDLE_PREV_NAME:-prev}.so *.o) 2>&1 | grep -i error
rm -rf $D
