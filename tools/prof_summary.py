#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV: per-step time by category + own kernels.

    python tools/prof_summary.py <kernel_stats.csv> <steps incl. warmup | auto> [--md]

`auto`: the number of optimizer steps in the trace (launches of this package's adamw_kernel, one per step).
"""
import csv
import sys


def cat(nm):
    if "fps_" in nm: return "FPS (hip)"
    if "ball_query" in nm or "grid_" in nm: return "ball_query+group (hip)"
    if "mha_" in nm: return "attention (hip)"
    if "coda" in nm and any(k in nm for k in ("sa_fwd_kernel", "sa_bwd_d", "dw_reduce", "pack_", "pool_finish", "pool_bwd_stats",
                                              "l1_sums", "l1_bwd")):
        return "SA shared-MLP MFMA pipeline (hip)"
    if "coda" in nm and any(k in nm for k in ("col_stats", "bn_relu_apply", "bn_bwd_sparse", "relu_bn")):
        return "SA shared-MLP streaming (hip)"
    if "coda" in nm and any(k in nm for k in ("bn_stats", "bn_finalize", "bn_act", "bn_bwd_finalize")):
        return "GenericMLP batch-norm kernels (hip)"
    if "coda" in nm and any(k in nm for k in ("add_ln", "colsum", "bias_relu_dropout")):
        return "transformer token kernels (hip)"
    if "coda" in nm and ("x3_nt" in nm or "x3_tn" in nm or "x3_split" in nm): return "own bf16x3 GEMMs, fp32-accurate (hip)"
    if "coda" in nm and ("sgemm" in nm or "grouped_tn" in nm or "gemm_tn" in nm): return "own fp32-MFMA GEMMs (hip)"
    if "coda" in nm and any(k in nm for k in ("giou", "hungarian", "box_decode", "box_loss", "align_loss", "nms", "box_point")):
        return "boxes / matcher / losses (hip)"
    if "coda" in nm and any(k in nm for k in ("adamw_kernel", "sumsq_kernel", "scale_kernel")): return "clip + AdamW (hip)"
    if "coda" in nm and any(k in nm for k in ("vit_attention", "ln_rows", "embed_ln", "patch_gather", "crop_resize", "project_rects")):
        return "CLIP image branch (hip)"
    if "coda" in nm: return "gather/group/interp (hip)"
    if "max_pool" in nm: return "max-pool (torch)"
    if "BatchNorm" in nm or "batch_norm" in nm: return "batch-norm (MIOpen)"
    if nm.startswith("Cijk") or "igemm" in nm or "miopenSp3" in nm: return "GEMM / conv (rocBLAS, MIOpen)"
    if "transpose" in nm: return "transpose (MIOpen)"
    if "softmax" in nm.lower(): return "softmax (torch)"
    if "dropout" in nm or "masked_scale" in nm: return "dropout (torch)"
    if "elementwise" in nm or "copy" in nm.lower() or "fill" in nm.lower(): return "elementwise / copy / fill (torch)"
    if "reduce" in nm: return "reductions (torch)"
    if "layer_norm" in nm.lower() or "LayerNorm" in nm or "GammaBeta" in nm: return "layer-norm (torch)"
    if "multi_tensor" in nm or "FusedAdam" in nm: return "optimizer (torch)"
    return "other"


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    if sys.argv[2] == "auto":
        n = sum(int(r["Calls"]) for r in rows if "adamw_kernel" in r["Name"])
    else:
        n = int(sys.argv[2])
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"GPU-busy per step: {tot / n / 1e6:.2f} ms; kernel launches per step: "
          f"{sum(int(r['Calls']) for r in rows) / n:.0f}\n")
    agg = {}
    for r in rows:
        agg[cat(r["Name"])] = agg.get(cat(r["Name"]), 0) + float(r["TotalDurationNs"])
    print("| category | ms/step | share |\n|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda x: -x[1]):
        print(f"| {k} | {v / n / 1e6:.3f} | {100 * v / tot:.1f}% |")
    print("\n| kernel (this repo) | calls/step | avg us | ms/step |\n|---|---|---|---|")
    for r in sorted([r for r in rows if "coda" in r["Name"]], key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"].replace("void coda::(anonymous namespace)::", "").replace("coda::(anonymous namespace)::", "")
        name = name.split("(")[0]
        print(f"| {name} | {int(r['Calls']) / n:.1f} | {float(r['AverageNs']) / 1e3:.1f} | "
              f"{float(r['TotalDurationNs']) / n / 1e6:.3f} |")


if __name__ == "__main__":
    main()
