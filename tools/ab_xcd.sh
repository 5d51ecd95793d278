#!/bin/bash
# dev: same-box A/B of the XCD-aware attention mapping + FETCH_SIZE of the new mapping
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_attention_gpu.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2; do
  echo "== XCD=0"; CODA_ATTN_XCD=0 python tools/bench_attn.py
  echo "== XCD=1"; CODA_ATTN_XCD=1 python tools/bench_attn.py
done
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/pmc_attn_FETCH_SIZE
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_attn_FETCH_SIZE -o run -- python $R/tools/bench_attn.py > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
c = "FETCH_SIZE"
for f in glob.glob(f"gpurun_out/pmc_attn_{c}/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            acc[(r["Kernel_Name"][40:110], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        v.sort()
        print(c, k, "n", len(v), "median", v[len(v) // 2], "min", v[0], "max", v[-1])
PY
