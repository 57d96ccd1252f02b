#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_multi_device.py tests/test_gpu_blob.py tests/test_c_abi.py -m gpu -x -q 2>&1 | tail -15
