#!/bin/bash
# dev: what the FIRST bench process of a fresh box reads against the box's later processes (the driver's order: smoke, bench)
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
show() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['ms_per_step'], 'unchanged', d.get('value_unchanged'), d['host'])"; }
CODA_BENCH_SETTLE_S=${FIRST_SETTLE_S:-6} python bench.py 2>/dev/null | show "first (default invocation)"
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | show "second"
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | show "third"
