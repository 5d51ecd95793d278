#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_sa_mfma_gpu.py tests/test_sa_module_gpu.py -q -m gpu -x 2>&1 | tail -2
timeout 300 python tools/sa_prof.py 2>&1 | grep -v amdgpu
timeout 300 python bench.py --workload sa --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('sa', d['value'], d['ms_per_step'])"
