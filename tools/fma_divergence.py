#!/usr/bin/env python3
"""How often do the three distance-arithmetic modes (include/coda_pointnet2.h) give different
indices?  CPU only (the oracle, which the HIP kernels match bit-exactly in every mode):
FPS 20000 -> 2048 and ball_query (r=0.2, 64 samples) on the synthetic bench scenes.

    python tools/fma_divergence.py [n_scenes=256]

Prints a markdown table; DESIGN.md section 2 quotes the result.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pointnet2_oracle as O  # noqa: E402
from coda_neurips2023_amd.synthetic_scenes import make_batch  # noqa: E402


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n, m, r, s = 20000, 2048, 0.2, 64
    B = 8
    stats = {k: dict(scenes=0, fps_idx=0, first=[], bq_rows=0, bq_rows_same_centres=0) for k in (1, 2)}
    total_rows = 0
    for b0 in range(0, n_scenes, B):
        pc, _, _ = make_batch(B, n, seed=1234 + b0)
        fps, bq0 = {}, None
        for mode in (0, 1, 2):
            O.set_fma_mode(mode)
            fps[mode] = O.furthest_point_sampling(pc, m)
        ctr0 = np.take_along_axis(pc, fps[0][..., None].astype(np.int64).repeat(3, -1), 1)
        for mode in (0, 1, 2):
            O.set_fma_mode(mode)
            bq = O.ball_query(ctr0, pc, r, s)  # same centres in every mode: isolates the ball test
            if mode == 0:
                bq0 = bq
                total_rows += bq.shape[0] * bq.shape[1]
            else:
                stats[mode]["bq_rows_same_centres"] += int((bq != bq0).any(-1).sum())
        for mode in (1, 2):
            diff = fps[mode] != fps[0]
            per_scene = diff.sum(1)
            stats[mode]["scenes"] += int((per_scene > 0).sum())
            stats[mode]["fps_idx"] += int(diff.sum())
            for row in diff:
                if row.any():
                    stats[mode]["first"].append(int(np.argmax(row)))
        print(f"  .. {b0 + B}/{n_scenes} scenes", file=sys.stderr)
    O.set_fma_mode(O.DEFAULT_FMA_MODE)
    print(f"| mode vs 0 | scenes with any FPS index change (of {n_scenes}) | changed FPS indices (of {n_scenes * m}) "
          f"| median first divergent sample | ball_query rows changed, same centres (of {total_rows}) |")
    print("|---|---|---|---|---|")
    for mode in (1, 2):
        st = stats[mode]
        first = int(np.median(st["first"])) if st["first"] else -1
        print(f"| {mode} | {st['scenes']} | {st['fps_idx']} | {first} | {st['bq_rows_same_centres']} |")


if __name__ == "__main__":
    main()
