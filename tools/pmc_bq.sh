#!/bin/bash
# dev: FETCH_SIZE / WRITE_SIZE of the ball-query operator (separate passes, --kernel-trace only), medians to stdout
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o bq -- python $R/tools/bench_ops.py bqonly > /dev/null 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  mkdir -p $R/gpurun_out/${PMC_TAG:-r05}_pmc
  cp "$f" $R/gpurun_out/${PMC_TAG:-r05}_pmc/bq_$c.csv
  python $R/tools/pmc_medians.py "$f" grid_
done
