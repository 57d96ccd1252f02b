#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_find_forms.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_real_text.py tests/test_gpu_configs.py tests/test_gpu_packed.py tests/test_reference_asserts.py tests/test_gpu_matches_txt_batch.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20
for rep in 1 2; do for k in 1 0; do echo "== NEEDLE_FIND_LENGTHS_PAIR=$k"; NEEDLE_FIND_LENGTHS_PAIR=$k python scripts/pair_shape_ab.py 2>&1 | grep -E "names7|kw12"; NEEDLE_FIND_LENGTHS_PAIR=$k python scripts/find_forms_ab.py 2>&1 | grep -E "names7|kw12"; done; done
timeout 900 python scripts/fuzz_campaign.py 30000 150 2>&1 | tail -1
