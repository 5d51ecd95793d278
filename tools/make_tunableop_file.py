#!/usr/bin/env python3
"""Builds coda_neurips2023_amd/tunableop_gfx950.csv, the table of library-GEMM winners that tuning.py loads, from
TunableOp result files of tuning runs on an MI355X:

    CODA_TUNED_GEMMS=0 PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=50 \
        PYTORCH_TUNABLEOP_FILENAME=$PWD/gpurun_out/tunable2/all.csv python bench.py --no-cpu-baseline
    python tools/make_tunableop_file.py gpurun_out/tunable2/all0.csv [more.csv ...]

Shapes with a fixed size (encoder / decoder / head projections) are copied.  The set-abstraction MLP runs on the
de-duplicated rows of a batch, a data-dependent count P that fused_sa_mlp.compact_groups rounds up to a multiple of
16 384: its products -- (P x 64) @ (64 x 128), (P x 128) @ (128 x 256), their input gradients, and the weight
gradients as P / 16 384 batched chunks -- were tuned for the P values the run met; the table carries every multiple of
16 384 from 131 072 to 1 310 720 with the winner of the NEAREST measured P (a winner is a tile configuration of the
library; between neighbouring row counts of a tall-skinny product it does not change what is measured).
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "coda_neurips2023_amd", "tunableop_gfx950.csv")
STEP, LO, HI = 16384, 131072, 1310720


def main():
    validators, fixed, fam_rows, fam_batch = [], {}, {}, {}
    for path in sys.argv[1:]:
        for line in open(path):
            line = line.strip()
            if not line:
                continue
            parts = line.split(",")
            if parts[0] == "Validator":
                if line not in validators:
                    validators.append(line)
                continue
            op, key, sol = parts[0], parts[1], parts[2]
            m = re.match(r"^([a-z]{2})_(\d+)_(\d+)_(\d+)(_B_(\d+))?_ld_(\d+)_(\d+)_(\d+)$", key)
            if not m:
                fixed[(op, key)] = sol
                continue
            tr, a, b, c, _, nb, l1, l2, l3 = m.groups()
            a, b, c = int(a), int(b), int(c)
            if nb is None and b >= LO and b % STEP == 0:          # rows in the second size field
                fam_rows.setdefault((op, tr, a, c, l1, l2, l3), {})[b] = sol
            elif nb is not None and c == STEP and a <= 256 and b <= 256:  # weight gradients: P / 16384 chunks of 16384 rows
                fam_batch.setdefault((op, tr, a, b, c, l1, l2, l3), {})[int(nb)] = sol
            else:
                fixed[(op, key)] = sol
    lines = list(validators)
    for (op, key), sol in sorted(fixed.items()):
        lines.append(f"{op},{key},{sol},0")
    nearest = lambda table, v: table[min(table, key=lambda k: (abs(k - v), k))]
    for (op, tr, a, c, l1, l2, l3), table in sorted(fam_rows.items()):
        for p in range(LO, HI + 1, STEP):
            lines.append(f"{op},{tr}_{a}_{p}_{c}_ld_{l1}_{l2}_{l3},{nearest(table, p)},0")
    for (op, tr, a, b, c, l1, l2, l3), table in sorted(fam_batch.items()):
        for nb in range(LO // STEP, HI // STEP + 1):
            lines.append(f"{op},{tr}_{a}_{b}_{c}_B_{nb}_ld_{l1}_{l2}_{l3},{nearest(table, nb)},0")
    with open(OUT, "w") as f:
        f.write("\n".join(lines) + "\n")
    print(f"{OUT}: {len(validators)} validators, {len(fixed)} fixed shapes, {len(fam_rows)} row families, "
          f"{len(fam_batch)} chunk families, {len(lines)} lines")


if __name__ == "__main__":
    main()
