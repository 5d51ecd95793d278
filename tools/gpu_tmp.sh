#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_full_step_gpu.py -q -m gpu -k configs4 2>&1 | grep -E "^E  |configs4|Error" | head -30
echo ---- with CODA_DEFER_SUMS=0
CODA_DEFER_SUMS=0 timeout 1200 python -m pytest tests/test_full_step_gpu.py -q -m gpu -k configs4 2>&1 | grep -E "^E  |configs4.*gradient|passed|failed" | head -10
