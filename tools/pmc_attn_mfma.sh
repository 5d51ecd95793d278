#!/bin/bash
# dev: MFMA-busy and wave-state counters of the fp32 attention kernels (one counter-collection run, --kernel-trace only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d /tmp/pmc_mfma -o attn -- python $R/tools/bench_attn.py fp32 > /dev/null 2>&1
f=$(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1)
mkdir -p $R/gpurun_out/r3_pmc
python - "$f" <<'PY'
import csv, statistics, sys
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list)); durs = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "mha_" not in n: continue
    k = n.replace("void ", "").replace("coda::(anonymous namespace)::", "").split("(")[0] + " grid " + r.get("Grid_Size", "")
    vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    durs[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel | dur us | MFMA busy (of 1024 SIMDs x 2.4 GHz) | parked WAIT_ANY | issue stall WAIT_INST_ANY | issuing ACTIVE_INST_ANY | VALU | LDS stall")
for k in sorted(vals):
    v = {c: statistics.median(x) for c, x in vals[k].items()}
    d = statistics.median(durs[k])
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    f = lambda c: 100 * v.get(c, 0) / wc
    print(f"{k:72s} | {d:7.1f} | {100 * v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (d * 1e-6 * 2.4e9 * 1024):5.1f} % | {f('SQ_WAIT_ANY'):4.0f} % | {f('SQ_WAIT_INST_ANY'):4.0f} % | {f('SQ_ACTIVE_INST_ANY'):4.0f} % | {f('SQ_ACTIVE_INST_VALU'):4.0f} % | {f('SQ_WAIT_INST_LDS'):3.0f} %")
PY
