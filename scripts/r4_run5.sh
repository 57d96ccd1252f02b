#!/bin/bash
# round 4: find-all with the compressed lengths program (MODE_SPARSE in find_all_kernel): parity tests, then C3-sparse find-all timing
# (int32 and one-dword forms) with the oracle check on sampled rows, against NEEDLE_FIND_LENGTHS_SPARSE=0 (hot rows + backward walks)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_find_all.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r4/tests5.log 2>&1; grep -E "passed|failed|^E  " gpurun_out/r4/tests5.log | tail -8
python -m pytest tests -x -q -m gpu -k "random_dictionaries or dictionar" >> gpurun_out/r4/tests5.log 2>&1; grep -E "passed|failed|^E  " gpurun_out/r4/tests5.log | tail -4
for e in "" "NEEDLE_FIND_LENGTHS_SPARSE=0"; do
  echo "== c3s find-all $e"
  env $e python scripts/find_all_probe.py c3s 10000000 32 check 2>&1 | grep -v amdgpu | tail -2
  env $e FIND_ALL_PROBE_PACKED=1 python scripts/find_all_probe.py c3s 10000000 32 check 2>&1 | grep -v amdgpu | tail -2
done | tee gpurun_out/r4/find_all_c3s_ab.log
