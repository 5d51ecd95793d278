#!/usr/bin/env python3
"""BASELINE.md section 3 part 1: configs[0] -- the REFERENCE's own Python plumbing (imported read-only from
/root/reference, as tests/golden/make_golden.py does) with the C oracle as ``pointnet2._ext``, on this container's
host cores: one SUN RGB-D-shaped scene, 20 000 points, 256 queries, forward only, eval mode, ``torch.no_grad``.
Runs only in the build container.  Prints per-stage and total medians (2 warm-ups, 5 timed runs)."""
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden import make_golden as MG  # noqa: E402
from oracle import pointnet2_oracle as O  # noqa: E402
from coda_neurips2023_amd.synthetic_scenes import make_batch  # noqa: E402


def main():
    threads = int(os.environ.get("CODA_CPU_THREADS", "8"))
    O.build()
    O.set_fma_mode(O.DEFAULT_FMA_MODE)
    MG.install_reference()
    torch.set_num_threads(threads)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    import models.model_3detr as M
    from datasets.sunrgbd_anonymous_aligned_image import SunrgbdAnonymousAlignedImageDatasetConfig as Cfg
    # main.py's defaults (main.py:64-74,127-148) = BASELINE.json's "model_3detr" configuration
    args = MG._args(nqueries=256, preenc_npoints=2048, enc_ffn_dim=128, enc_nlayers=3, enc_dropout=0.1, dec_dim=256,
                    dec_ffn_dim=256, dec_nlayers=8, dec_dropout=0.1, mlp_dropout=0.3)
    cfg = Cfg(if_print=False, args=args)
    pre, enc, dec = M.build_preencoder(args), M.build_encoder(args), M.build_decoder(args)
    model = M.Model3DETRPredictedBoxDistillationHead(pre, enc, dec, cfg, encoder_dim=256, decoder_dim=args.dec_dim,
                                                     mlp_dropout=0.3, num_queries=256, if_with_clip_train=False, args=args)
    model.eval()
    pc, mn, mx = make_batch(1, 20000, seed=4242)
    pcs, dims = torch.from_numpy(pc), [torch.from_numpy(mn), torch.from_numpy(mx)]
    stages = {}

    def timed(name, fn):
        t0 = time.perf_counter()
        out = fn()
        stages.setdefault(name, []).append(time.perf_counter() - t0)
        return out

    def forward():
        with torch.no_grad():
            enc_xyz, enc_feat, _ = timed("run_encoder (SA stage + 3 encoder layers)", lambda: model.run_encoder(pcs))
            enc_feat = timed("encoder_to_decoder_projection",
                             lambda: model.encoder_to_decoder_projection(enc_feat.permute(1, 2, 0)).permute(2, 0, 1))
            q_xyz, q_emb = timed("get_query_embeddings (FPS 2048->256 + MLP)", lambda: model.get_query_embeddings(enc_xyz, dims))
            pos = model.pos_embedding(enc_xyz, input_range=dims).permute(2, 0, 1)
            q_emb = q_emb.permute(2, 0, 1)
            box = timed("decoder (8 layers)", lambda: model.decoder(torch.zeros_like(q_emb), enc_feat, query_pos=q_emb, pos=pos)[0])
            timed("get_box_predictions (6 heads + decode)", lambda: model.get_box_predictions(q_xyz, dims, box, pcs, {}))

    totals = []
    for it in range(7):
        stages_before = {k: len(v) for k, v in stages.items()}
        t0 = time.perf_counter()
        forward()
        dt = time.perf_counter() - t0
        if it >= 2:
            totals.append(dt)
        else:  # warm-up: drop its stage samples
            for k in stages:
                del stages[k][stages_before.get(k, 0):]
    print(f"threads {threads}; torch {torch.__version__}; 1 scene, 20000 points, 256 queries, eval, no_grad")
    for k, v in stages.items():
        print(f"  {k:52s} {np.median(v) * 1e3:8.1f} ms")
    print(f"  total forward: median {np.median(totals):.3f} s ({1.0 / np.median(totals):.2f} scenes/s), min {min(totals):.3f} s")


if __name__ == "__main__":
    main()
