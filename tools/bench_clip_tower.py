"""Frozen CLIP image tower (ViT-B/16, the reference's, models/model_3detr.py:325) on the crops of one training
step: 8 scenes x 32 proposals = 256 crops (models/model_3detr.py:991).  Prints crops/s, ms per call and the
matrix-core rate (algorithmic flops: GEMMs + attention products) against the dense fp16 MFMA peak.

    python tools/bench_clip_tower.py [--crops 256] [--dtype fp16|fp32] [--iters 10]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coda_neurips2023_amd import _lib, clip_tower  # noqa: E402

PEAK = {"fp16": 2500.0, "fp32": 157.0}  # TFLOP/s dense, MI355X_MICROARCH.md


def tower_flops(res=224, patch=16, width=768, layers=12, mlp=3072, out=512):
    g = res // patch
    l = g * g + 1
    per_layer = 2 * l * width * (3 * width + width + 2 * mlp) + 4 * l * l * width
    return layers * per_layer + 2 * g * g * 3 * patch * patch * width + 2 * width * out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crops", type=int, default=256)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    torch.manual_seed(0)
    tower = clip_tower.ImageTower(512, 224, 12, 768, 16)
    tower = (clip_tower.convert_weights(tower) if a.dtype == "fp16" else tower).cuda()
    x = torch.randn(a.crops, 3, 224, 224, device="cuda")
    for _ in range(3):
        tower.encode_image(x)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
    ev[0].record()
    for i in range(a.iters):
        tower.encode_image(x)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters))
    med = ms[len(ms) // 2]
    tf = tower_flops() * a.crops / (med * 1e-3) / 1e12
    print(json.dumps({"what": "clip_image_tower ViT-B/16", "dtype": a.dtype, "crops": a.crops, "ms_per_call": round(med, 3),
                      "min_ms": round(ms[0], 3), "crops_per_s": round(a.crops / (med * 1e-3), 1),
                      "gflop_per_crop": round(tower_flops() / 1e9, 2), "tflops": round(tf, 1),
                      "frac_of_mfma_peak": round(tf / PEAK[a.dtype], 3),
                      "quickgelu_in_gemm_epilogue": _lib.load().coda_vit_quickgelu_fused(1 if a.dtype == "fp16" else 0)}))


if __name__ == "__main__":
    main()
