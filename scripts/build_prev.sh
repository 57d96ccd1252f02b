#!/bin/bash
# builds the library as of git revision $1 (default HEAD) into needle_amd/libneedle_hip_prev.so for scripts/ab.sh
REV=${1:-HEAD}
D=$(mktemp -d)
mkdir -p $D/needle_amd/csrc $D/include
for f in needle_kernels.hip needle_api.cpp needle_lower.cpp needle_regex.cpp needle_device.h needle_lower.h needle_regex.h; do git show $REV:needle_amd/csrc/$f > $D/needle_amd/csrc/$f; done
git show $REV:include/needle_hip.h > $D/include/needle_hip.h
(cd $D/needle_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -x hip -Wno-unused-variable -o /root/repo/needle_amd/libneedle_hip_prev.so needle_kernels.hip needle_api.cpp needle_lower.cpp needle_regex.cpp) 2>&1 | grep -i error
rm -rf $D
