#!/bin/bash
# dev: FETCH_SIZE / WRITE_SIZE of the fp32 attention kernels on the model's shapes (separate passes, --kernel-trace only;
# FETCH_SIZE is reported in KB and counts half the bytes on gfx950: x 2), per-kernel medians to stdout
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o attn -- python $R/tools/bench_attn.py fp32 > /dev/null 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  mkdir -p $R/gpurun_out/${PMC_TAG:-r05}_pmc
  cp "$f" $R/gpurun_out/${PMC_TAG:-r05}_pmc/attn_$c.csv
  echo "== $c"
  python $R/tools/pmc_medians.py "$f" mha_
done
