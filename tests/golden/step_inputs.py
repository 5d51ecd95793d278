"""Seeded inputs of the full-size whole-step fixture (tests/golden/step_full.npz): the SAME scenes, targets and
region-embedding seam tensors tests/test_full_step_gpu.py uses for configs[2].  Shared by the generator (which feeds
them to the REFERENCE's modules) and the test (which feeds them to this package): the fixture carries outputs only."""
import math

import torch
import torch.nn.functional as F

B, NCLS = 8, 10
# the per-GPU shares of BASELINE.json's configs: points per scene, queries, decoder width, loss recipe
# (stage 1 = scripts/coda_sunrgbd_stage1.sh:7-27 at its own shape: d_dec 512, 128 queries, L1 alignment term only)
CASES = {"configs2": dict(npoints=20000, nq=256, dec_dim=256, stage=2),
         "configs3": dict(npoints=20000, nq=128, dec_dim=512, stage=1)}
WEIGHT_SEED = 23
SAMPLES = 1024            # strided entries kept per tensor
LOGIT_SCALE = 100.0       # clip(exp(ln 100), max=100): models/model_3detr.py:1796 with a released CLIP checkpoint
LOGIT_SCALE_PARAM = math.log(100.0)


def recipe(case):
    """The argparse namespace of the case (bench.recipe_args: main.py's defaults + the stage's script), dropout 0."""
    import bench
    c = CASES[case]
    args = bench.recipe_args(c["nq"], dec_dim=c["dec_dim"], enc_dropout=0.0, dec_dropout=0.0, mlp_dropout=0.0)
    if c["stage"] == 1:
        args.loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi_weight = 0
    return args


def build(case):
    """-> (batch dict of CPU tensors incl. the ground-truth entries, seam dict)."""
    NQ, NPOINTS = CASES[case]["nq"], CASES[case]["npoints"]
    import bench
    from coda_neurips2023_amd.synthetic_scenes import make_batch
    gen = torch.Generator().manual_seed(11)
    seam = {"text": F.normalize(torch.randn(NCLS, 512, generator=gen), dim=-1),
            "img_emb": F.normalize(torch.randn(B, NQ, 512, generator=gen), dim=-1),
            "mask": (torch.rand(B, NQ, 1, generator=gen) < 0.25).float(),
            "weak_label": torch.randint(0, NCLS, (B, NQ), generator=gen),
            "weak_conf": torch.rand(B, NQ, generator=gen) * (torch.rand(B, NQ, generator=gen) < 0.5)}
    pc, mn, mx = make_batch(B, NPOINTS, seed=555)
    batch = {"point_clouds": torch.from_numpy(pc), "point_cloud_dims_min": torch.from_numpy(mn),
             "point_cloud_dims_max": torch.from_numpy(mx)}
    batch.update(bench.synthetic_targets(batch, torch.Generator().manual_seed(2)))
    return batch, seam
