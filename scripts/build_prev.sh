#!/bin/bash
# builds the library as of git revision $1 (default HEAD) into needle_amd/libneedle_hip_prev.so for scripts/ab.sh
REV=${1:-HEAD}
D=$(mktemp -d)
git archive $REV needle_amd/csrc include | tar -x -C $D
(cd $D/needle_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -x hip -Wno-unused-variable -o /root/repo/needle_amd/libneedle_hip_prev.so $(ls *.hip *.cpp | grep -v stream_probe)) 2>&1 | grep -i error
rm -rf $D
