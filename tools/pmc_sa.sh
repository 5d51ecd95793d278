#!/bin/bash
# dev: PMC counters of the set-abstraction MFMA kernels (counter collection with --kernel-trace only), two passes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
rm -rf /tmp/pmc_sa1 /tmp/pmc_sa2 /tmp/pmc_sa3 /tmp/pmc_sa4
CMD="python $R/bench.py --workload sa --steps 6 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d /tmp/pmc_sa1 -o sa -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU \
  --kernel-trace --output-format csv -d /tmp/pmc_sa2 -o sa -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_sa3 -o sa -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_sa4 -o sa -- $CMD > /dev/null 2>&1
python - $(find /tmp/pmc_sa1 /tmp/pmc_sa2 /tmp/pmc_sa3 /tmp/pmc_sa4 -name "*counter_collection.csv") > $R/gpurun_out/pmc_sa.txt <<'PY'
import csv, statistics, sys
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list)); durs = defaultdict(list)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "sa_" not in n and "pack_" not in n and "dw_reduce" not in n and "pool_" not in n: continue
        k = n.replace("void ", "").replace("coda::(anonymous namespace)::", "").split("(")[0]
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        durs[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(vals):
    v = {c: statistics.median(x) for c, x in vals[k].items()}
    d = statistics.median(durs[k])
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{k}: dur {d:.1f} us")
    print("   MFMA busy %.1f %% of (dur x 2.4 GHz x 1024 SIMDs); wave-cycle shares: WAIT_ANY %.0f%% WAIT_INST_ANY %.0f%% ACTIVE_INST_ANY %.0f%% VALU %.0f%% WAIT_INST_LDS %.0f%%; BUSY_CYCLES %.3g" % (
        100 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (d * 1e-6 * 2.4e9 * 1024), 100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc,
        100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * v.get("SQ_WAIT_INST_LDS", 0) / wc, v.get("SQ_BUSY_CYCLES", 0)))
    print("   insts: VALU %.3g MFMA %.3g LDS %.3g VMEM_RD %.3g VMEM_WR %.3g SALU %.3g; LDS bank conflict cycles %.3g of %.3g active" % (
        v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_MFMA", 0), v.get("SQ_INSTS_LDS", 0), v.get("SQ_INSTS_VMEM_RD", 0), v.get("SQ_INSTS_VMEM_WR", 0), v.get("SQ_INSTS_SALU", 0),
        v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_LDS_IDX_ACTIVE", 0)))
    print("   FETCH_SIZE %.4g KB (x2 on gfx950 per the guide) WRITE_SIZE %.4g KB" % (v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)))
PY
cat $R/gpurun_out/pmc_sa.txt
