#!/bin/bash
# dev: same-box A/B of one environment switch on the whole step:  ab_step.sh VAR REPS STEPS VAL...
# (VAL "-" = variable unset); prints value / ms_per_step / value_unchanged per run
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=$1; REPS=$2; STEPS=$3; shift 3
for r in $(seq $REPS); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then E="env -u $VAR"; else E="env $VAR=$v"; fi
    $E python bench.py --no-cpu-baseline --no-extras --steps $STEPS --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', d['value'], d['ms_per_step'], d.get('config', {}).get('value_unchanged'), d['roofline'].get('avg_launch_ms'))"
  done
done
