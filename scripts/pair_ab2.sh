#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rx in "sherlock|holmes|watson|irene|adler|john|baker" "http://.+"; do
  for lib in libneedle_hip_prev.so libneedle_hip.so libneedle_hip_tf.so; do
    echo "== $lib"; NEEDLE_LIB=$PWD/needle_amd/$lib python scripts/quick_regex.py c3 "$rx" 2>&1 | grep -v "amdgpu\|matches "
  done
done
bash scripts/ab.sh "needle_amd/libneedle_hip_tf.so" c3
