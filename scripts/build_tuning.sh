#!/bin/bash
# builds the working tree with -DNEEDLE_TUNING (measurement-only switches such as NEEDLE_DEBUG_NO_BACKWARD, which change
# ANSWERS and are therefore compiled out of the shipping library) into needle_amd/libneedle_hip_tuning.so; use it through
# NEEDLE_LIB=$PWD/needle_amd/libneedle_hip_tuning.so (scripts/ab.sh, scripts/find_all_probe.py)
cd "$(dirname "$0")/../needle_amd/csrc" || exit 1
D=$(mktemp -d)
for f in $(ls *.hip *.cpp | grep -v stream_probe); do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-variable -DNEEDLE_TUNING $NEEDLE_EXTRA_DEFS -c $f -o $D/$f.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libneedle_hip_tuning${NEEDLE_TUNING_SUFFIX}.so $D/*.o
rm -rf $D
