#!/bin/bash
# dev: one GPU-box call that refreshes the round's measured artifacts under gpurun_out/
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_attention_gpu.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -3
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
tail -c 3000 gpurun_out/final_bench.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_attn_$c
  timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_attn_$c -o run -- python $R/tools/bench_attn.py > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_attn_{c}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[(r["Kernel_Name"][:60], r["Grid_Size"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            if "mha" in k[0]:
                v.sort()
                print(c, k, "n", len(v), "median", v[len(v) // 2], "min", v[0], "max", v[-1])
PY
