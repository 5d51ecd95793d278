#!/bin/bash
# dev: the headline leg N times in a row on one box (value, ms/step, value_unchanged, host enqueue ms, dominant kernel ms)
cd "$GRAFT_REPO_ROOT" || exit 1
N=${1:-6}
for i in $(seq $N); do
  python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('run $i:', d['value'], 'scenes/s', d['ms_per_step'], 'ms  unchanged', d.get('value_unchanged'), ' host enqueue', d['host']['enqueue_ms_per_step'], ' slack', d['host']['slack_probe_ms'], ' dK/dV ms', d['roofline'].get('avg_launch_ms'), ' decoder attn', d['roofline']['north_star'].get('decoder_attention_frac_algorithmic'))"
done
