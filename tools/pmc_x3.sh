#!/bin/bash
# dev: PMC counters of the bf16x3 GEMM kernels (counter collection with --kernel-trace only, one counter group per pass) on
# tools/x3_dbg.py's three shapes -> gpurun_out/pmc_x3.txt
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
rm -rf /tmp/pmc_x31 /tmp/pmc_x32 /tmp/pmc_x33 /tmp/pmc_x34
CMD="python $R/tools/x3_dbg.py"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d /tmp/pmc_x31 -o x3 -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU \
  --kernel-trace --output-format csv -d /tmp/pmc_x32 -o x3 -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_x33 -o x3 -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_x34 -o x3 -- $CMD > /dev/null 2>&1
python - $(find /tmp/pmc_x31 /tmp/pmc_x32 /tmp/pmc_x33 /tmp/pmc_x34 -name "*counter_collection.csv") > $R/gpurun_out/pmc_x3.txt <<'PY'
import csv, statistics, sys
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list)); durs = defaultdict(list)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "x3_" not in n: continue
        k = n.replace("void ", "").replace("coda::(anonymous namespace)::", "").split("(")[0] + " grid " + r.get("Grid_Size", "?")
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        durs[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("tools/x3_dbg.py launches, in order of first appearance: 98304 x 256 x 256, 16384 x 2048 x 256, 16384 x 256 x 2048 (all grid 131072 = 256 workgroups x 512);")
print("one kernel name + grid serves all three, so the medians below mix them -- read the per-shape rows of the second block")
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
# per shape: the launches of a pass come in blocks of 23 per shape (3 warm-up + 20 timed), in the order above
print()
for f in sys.argv[1:]:
    rows = [r for r in csv.DictReader(open(f)) if "x3_nt" in r["Kernel_Name"]]
    by_counter = defaultdict(list)
    for r in rows:
        by_counter[r["Counter_Name"]].append(r)
    for c, rs in by_counter.items():
        rs.sort(key=lambda r: int(r["Start_Timestamp"]))
        n = len(rs) // 3
        for i, shape in enumerate(("98304x256x256", "16384x2048x256", "16384x256x2048")):
            blk = rs[i * n:(i + 1) * n]
            if blk:
                med = statistics.median(float(r["Counter_Value"]) for r in blk)
                dur = statistics.median((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in blk)
                print(f"{shape:16s} {c:28s} median {med:14.1f}   (dur {dur:7.1f} us, {len(blk)} launches)")
PY
cat $R/gpurun_out/pmc_x3.txt
