#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ FROM THE REFERENCE.

Runs only in the build container (needs /root/reference, read-only).  It
imports the reference's own Python layers -- third_party_pointnet2/pointnet2/
{pointnet2_utils,pointnet2_modules,pytorch_utils}.py, models/{transformer,
helpers,position_embedding,model_3detr}.py, criterion.py -- on CPU torch, with
the CPU oracle (oracle/pointnet2_oracle.c) registered as ``pointnet2._ext``
(the reference's CUDA ops have no CPU path and there is no nvcc here), and
stores inputs / seeded weights / outputs / gradients as .npz.

Nothing of the reference's source text is stored: the fixtures are arrays only.

    python tests/golden/make_golden.py            # regenerate everything
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import pointnet2_oracle as O  # noqa: E402
from coda_neurips2023_amd.synthetic_scenes import make_batch  # noqa: E402


class _Stub(types.ModuleType):
    """Module whose every attribute is a harmless callable / namespace."""

    def __init__(self, name):
        super().__init__(name)
        self.__path__ = []

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Anything()


class _Anything:
    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Anything()


def install_reference(any_dtype=False):
    """stub the third-party imports that are absent here and
    register the oracle as pointnet2._ext BEFORE importing the reference.  any_dtype: the operators follow the tensors'
    dtype (oracle/cpu_port._AnyDtypeExt: index-producing operators on the float32 image of their input, gathers as
    torch indexing) -- for the float64 run of golden_step_full_f64."""
    for name in ["plyfile", "trimesh", "cv2", "ftfy", "torchvision", "torchvision.transforms",
                 "torchvision.ops", "torchvision.models", "torchvision.models.detection",
                 "torchvision.models.detection.backbone_utils", "timm", "timm.data",
                 "timm.data.constants", "models.vision_transformer", "models.resnet",
                 "tensorboardX"]:
        if name not in sys.modules:
            sys.modules[name] = _Stub(name)
    const = sys.modules["timm.data.constants"]
    const.IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
    const.IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
    const.DEFAULT_CROP_PCT = 0.875
    pkg = types.ModuleType("pointnet2")
    pkg.__path__ = []
    if any_dtype:
        from oracle.cpu_port import _AnyDtypeExt
        ext = _AnyDtypeExt()
    else:
        ext = O.TorchExt()
    mod = types.ModuleType("pointnet2._ext")
    for fn in ["gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn",
               "three_interpolate", "three_interpolate_grad", "ball_query", "group_points",
               "group_points_grad"]:
        setattr(mod, fn, getattr(ext, fn))
    pkg._ext = mod
    sys.modules["pointnet2"] = pkg
    sys.modules["pointnet2._ext"] = mod
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "third_party_pointnet2", "pointnet2"))
    os.chdir(REF)


def _np(t):
    return t.detach().cpu().numpy()


FMA_MODE = O.DEFAULT_FMA_MODE  # distance-arithmetic mode of the oracle ops while generating


def _save(name, **arrays):
    arrays["fma_mode"] = np.int32(O.get_fma_mode())
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# --------------------------------------------------------------------------------------
def golden_ops():
    """The nine ops through the reference's autograd Functions (pointnet2_utils.py)."""
    import pointnet2_utils as RU  # the REFERENCE module

    out = {}
    cases = [("small", 2, 9, 2, 3, 5.0, 4), ("mid", 2, 1024, 128, 64, 0.3, 6)]
    for tag, b, n, m, s, radius, c in cases:
        g = torch.Generator().manual_seed(100 + n)
        if n == 9:  # the reference's smoke shapes (pointnet2_modules.py:497-499)
            xyz = torch.randn(b, n, 3, generator=g)
        else:
            pc, _, _ = make_batch(b, n, seed=4321)
            xyz = torch.from_numpy(pc)
        feats = torch.randn(b, c, n, generator=g).requires_grad_(True)
        inds = RU.furthest_point_sample(xyz, m)
        new_xyz = RU.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
        idx = RU.ball_query(radius, s, xyz, new_xyz)
        grouped = RU.grouping_operation(feats, idx)
        gw = torch.randn(grouped.shape, generator=g)
        (grouped * gw).sum().backward()
        gfeats = feats.grad.clone()
        feats.grad = None
        gathered = RU.gather_operation(feats, inds)
        gw2 = torch.randn(gathered.shape, generator=g)
        (gathered * gw2).sum().backward()
        ggather = feats.grad.clone()
        # three_nn / three_interpolate: propagate features of the m centres back to the n points
        known_feats = torch.randn(b, c, m, generator=g).requires_grad_(True)
        dist, nn_idx = RU.three_nn(xyz, new_xyz)
        dist_recip = 1.0 / (dist + 1e-8)
        weight = dist_recip / dist_recip.sum(2, keepdim=True)
        interp = RU.three_interpolate(known_feats, nn_idx, weight)
        gw3 = torch.randn(interp.shape, generator=g)
        (interp * gw3).sum().backward()
        out.update({
            f"{tag}_xyz": _np(xyz), f"{tag}_feats": _np(feats), f"{tag}_radius": np.float32(radius),
            f"{tag}_nsample": np.int32(s), f"{tag}_fps": _np(inds), f"{tag}_new_xyz": _np(new_xyz),
            f"{tag}_ball_idx": _np(idx), f"{tag}_grouped": _np(grouped), f"{tag}_group_gw": _np(gw),
            f"{tag}_group_grad": _np(gfeats), f"{tag}_gathered": _np(gathered),
            f"{tag}_gather_gw": _np(gw2), f"{tag}_gather_grad": _np(ggather),
            f"{tag}_known_feats": _np(known_feats), f"{tag}_nn_dist": _np(dist),
            f"{tag}_nn_idx": _np(nn_idx), f"{tag}_nn_weight": _np(weight), f"{tag}_interp": _np(interp),
            f"{tag}_interp_gw": _np(gw3), f"{tag}_interp_grad": _np(known_feats.grad),
        })
    # known-answer constants of the reference's own test (pointnet2_test.py:15-30)
    g = torch.Generator().manual_seed(7)
    feats = torch.randn(1, 2, 4, generator=g).requires_grad_(True)
    idx = torch.from_numpy(np.array([[[0, 1, 2], [1, 2, 3]]])).int()
    weight = torch.from_numpy(np.array([[[1, 1, 1], [2, 2, 2]]])).float()
    interp = RU.three_interpolate(feats, idx, weight)
    interp.sum().backward()
    out.update({"kat_feats": _np(feats), "kat_idx": _np(idx), "kat_weight": _np(weight),
                "kat_interp": _np(interp), "kat_grad": _np(feats.grad)})
    _save("pointnet2_ops.npz", **out)


def golden_sa_module():
    """PointnetSAModuleVotes (pointnet2_modules.py:161-268) with seeded weights,
    train-mode BN (batch statistics) and eval-mode BN, outputs + weight grads."""
    import pointnet2_modules as RM  # the REFERENCE module

    out = {}
    for tag, use_feats in [("xyz", False), ("feat", True)]:
        torch.manual_seed(11)
        b, n, npoint, nsample, radius = 2, 1024, 128, 32, 0.3
        c_in = 5 if use_feats else 0
        mod = RM.PointnetSAModuleVotes(mlp=[c_in, 16, 32, 64], npoint=npoint, radius=radius,
                                       nsample=nsample, normalize_xyz=True)
        # non-trivial BN affine + running stats so eval mode is a real test
        with torch.no_grad():
            for k, p in mod.named_parameters():
                if "bn" in k:
                    p.copy_(torch.rand_like(p) + 0.5 if k.endswith("weight") else torch.randn_like(p) * 0.1)
        pc, _, _ = make_batch(b, n, seed=555)
        xyz = torch.from_numpy(pc)
        feats = torch.randn(b, c_in, n).requires_grad_(True) if use_feats else None
        state0 = {k: _np(v).copy() for k, v in mod.state_dict().items()}
        mod.train()
        new_xyz, new_feat, inds = mod(xyz, feats)
        gw = torch.randn_like(new_feat)
        (new_feat * gw).sum().backward()
        grads = {k: _np(p.grad) for k, p in mod.named_parameters()}
        state1 = {k: _np(v).copy() for k, v in mod.state_dict().items()}  # BN running stats moved
        mod.eval()
        with torch.no_grad():
            _, new_feat_eval, _ = mod(xyz, feats)
        out.update({f"{tag}_xyz": _np(xyz), f"{tag}_new_xyz": _np(new_xyz), f"{tag}_inds": _np(inds),
                    f"{tag}_new_feat_train": _np(new_feat), f"{tag}_gw": _np(gw),
                    f"{tag}_new_feat_eval": _np(new_feat_eval)})
        if use_feats:
            out[f"{tag}_feats"] = _np(feats)
            out[f"{tag}_feats_grad"] = _np(feats.grad)
        for k, v in state0.items():
            out[f"{tag}_state0/{k}"] = v
        for k, v in state1.items():
            out[f"{tag}_state1/{k}"] = v
        for k, v in grads.items():
            out[f"{tag}_grad/{k}"] = v
    _save("sa_module.npz", **out)


def _args(**over):
    from types import SimpleNamespace
    ns = SimpleNamespace(use_color=False, enc_dim=256, preenc_npoints=128, enc_type="vanilla", enc_nhead=4,
                         enc_ffn_dim=64, enc_dropout=0.0, enc_activation="relu", enc_nlayers=2, dec_dim=64,
                         dec_nhead=4, dec_ffn_dim=64, dec_dropout=0.0, dec_nlayers=3, mlp_dropout=0.0,
                         nqueries=32, dataset_name="sunrgbd", begin_keep_epoch=1,
                         online_nms_update_save_novel_label_clip_driven_with_cate_confidence=False,
                         save_objectness=0.3, online_nms_update_save_epoch=50, clip_driven_keep_thres=0.3,
                         eval_layer_id=-1, if_clip_weak_labels=False, if_accumulate_former_pseudo_labels=False,
                         if_use_v1=True, image_size_width=730, image_size_height=531, test_range_min=0,
                         test_range_max=10, train_range_min=0, train_range_max=10)
    for k, v in over.items():
        setattr(ns, k, v)
    return ns


def golden_sa_module_wide():
    """The pre-encoder's own widths: PointnetSAModuleVotes(mlp=[0, 64, 128, 256]) of the REFERENCE
    (pointnet2_modules.py:161-268, SharedMLP pytorch_utils.py:8-33), xyz only, train-mode batch statistics, a
    quarter of the BatchNorm gammas negative (the pooled layer then keeps a group's MINIMUM pre-BN value) --
    the fixture of the hand-written MFMA pipeline (csrc/sa_mfma.hip), which is instantiated for these widths only."""
    import pointnet2_modules as RM  # the REFERENCE module

    torch.manual_seed(21)
    b, n, npoint, nsample, radius = 2, 2048, 96, 64, 0.25
    mod = RM.PointnetSAModuleVotes(mlp=[0, 64, 128, 256], npoint=npoint, radius=radius, nsample=nsample,
                                   normalize_xyz=True)
    with torch.no_grad():
        for k, p in mod.named_parameters():
            if "bn" in k and k.endswith("weight"):
                p.copy_((torch.rand_like(p) + 0.5) * torch.where(torch.rand_like(p) < 0.25, -1.0, 1.0))
            elif "bn" in k:
                p.copy_(torch.randn_like(p) * 0.1)
    pc, _, _ = make_batch(b, n, seed=556)
    xyz = torch.from_numpy(pc)
    out = {"xyz": _np(xyz)}
    for k, v in mod.state_dict().items():
        out[f"state0/{k}"] = _np(v).copy()
    mod.train()
    new_xyz, new_feat, inds = mod(xyz, None)
    gw = torch.randn_like(new_feat)
    (new_feat * gw).sum().backward()
    out.update({"new_xyz": _np(new_xyz), "inds": _np(inds), "new_feat_train": _np(new_feat), "gw": _np(gw)})
    for k, p in mod.named_parameters():
        out[f"grad/{k}"] = _np(p.grad)
    for k, v in mod.state_dict().items():
        out[f"state1/{k}"] = _np(v).copy()
    mod.eval()
    with torch.no_grad():
        _, new_feat_eval, _ = mod(xyz, None)
    out["new_feat_eval"] = _np(new_feat_eval)
    _save("sa_module_wide.npz", **out)


def golden_transformer():
    """TransformerEncoder / MaskedTransformerEncoder / TransformerDecoder stacks
    (models/transformer.py) at d=64 with full gradient digests, plus single layers at
    the real width d=256, h=4.  dropout=0 so train-mode graphs are deterministic."""
    from golden.weights import fill_deterministic, grad_digest
    import models.transformer as RT  # the REFERENCE module
    import pointnet2_modules as RM

    out = {}
    g = torch.Generator().manual_seed(5)
    # --- encoder stack, d=64
    layer = RT.TransformerEncoderLayer(d_model=64, nhead=4, dim_feedforward=32, dropout=0.0)
    enc = fill_deterministic(RT.TransformerEncoder(layer, 3), seed=1).train()
    src = torch.randn(96, 2, 64, generator=g).requires_grad_(True)
    _, y, _ = enc(src)
    gw = torch.randn(y.shape, generator=g)
    (y * gw).sum().backward()
    out.update({"enc_src": _np(src), "enc_out": _np(y), "enc_gw": _np(gw), "enc_src_grad": _np(src.grad)})
    for k, v in grad_digest(enc).items():
        out[f"enc_grad/{k}"] = v
    # --- masked encoder with interim set-abstraction down-sampling, d=64
    pc, _, _ = make_batch(2, 64, seed=77)
    xyz = torch.from_numpy(pc)
    layer = RT.TransformerEncoderLayer(d_model=64, nhead=4, dim_feedforward=32, dropout=0.0)
    interim = RM.PointnetSAModuleVotes(radius=0.6, nsample=8, npoint=32, mlp=[64, 32, 64], normalize_xyz=True)
    menc = fill_deterministic(RT.MaskedTransformerEncoder(layer, 3, masking_radius=[0.8, 1.6, 2.4],
                                                          interim_downsampling=interim), seed=2).train()
    src = torch.randn(64, 2, 64, generator=g).requires_grad_(True)
    mxyz, my, minds = menc(src, xyz=xyz)
    gw = torch.randn(my.shape, generator=g)
    (my * gw).sum().backward()
    out.update({"menc_xyz_in": _np(xyz), "menc_src": _np(src), "menc_out": _np(my), "menc_xyz": _np(mxyz),
                "menc_inds": _np(minds), "menc_gw": _np(gw), "menc_src_grad": _np(src.grad)})
    for k, v in grad_digest(menc).items():
        out[f"menc_grad/{k}"] = v
    # --- decoder stack, d=64, 4 layers, intermediate outputs
    dl = RT.TransformerDecoderLayer(d_model=64, nhead=4, dim_feedforward=48, dropout=0.0)
    dec = fill_deterministic(RT.TransformerDecoder(dl, 4, return_intermediate=True), seed=3).train()
    memory = torch.randn(96, 2, 64, generator=g).requires_grad_(True)
    pos = torch.randn(96, 2, 64, generator=g)
    qpos = torch.randn(24, 2, 64, generator=g).requires_grad_(True)
    tgt = torch.zeros(24, 2, 64)
    y, attns = dec(tgt, memory, query_pos=qpos, pos=pos, return_attn_weights=True)
    gw = torch.randn(y.shape, generator=g)
    (y * gw).sum().backward()
    out.update({"dec_memory": _np(memory), "dec_pos": _np(pos), "dec_qpos": _np(qpos), "dec_out": _np(y),
                "dec_attns": _np(attns), "dec_gw": _np(gw), "dec_memory_grad": _np(memory.grad),
                "dec_qpos_grad": _np(qpos.grad)})
    for k, v in grad_digest(dec).items():
        out[f"dec_grad/{k}"] = v
    # --- single layers at the real width (d=256, h=4), outputs + input grads
    el = fill_deterministic(RT.TransformerEncoderLayer(d_model=256, nhead=4, dim_feedforward=128, dropout=0.0),
                            seed=4).train()
    src = torch.randn(160, 2, 256, generator=g).requires_grad_(True)
    y = el(src)
    gw = torch.randn(y.shape, generator=g)
    (y * gw).sum().backward()
    out.update({"el_src": _np(src), "el_out": _np(y), "el_gw": _np(gw), "el_src_grad": _np(src.grad)})
    dl = fill_deterministic(RT.TransformerDecoderLayer(d_model=256, nhead=4, dim_feedforward=256, dropout=0.0),
                            seed=5).train()
    tgt = torch.randn(40, 2, 256, generator=g).requires_grad_(True)
    memory = torch.randn(160, 2, 256, generator=g).requires_grad_(True)
    pos = torch.randn(160, 2, 256, generator=g)
    qpos = torch.randn(40, 2, 256, generator=g)
    y, _ = dl(tgt, memory, pos=pos, query_pos=qpos)
    gw = torch.randn(y.shape, generator=g)
    (y * gw).sum().backward()
    out.update({"dl_tgt": _np(tgt), "dl_memory": _np(memory), "dl_pos": _np(pos), "dl_qpos": _np(qpos),
                "dl_out": _np(y), "dl_gw": _np(gw), "dl_tgt_grad": _np(tgt.grad),
                "dl_memory_grad": _np(memory.grad)})
    _save("transformer.npz", **out)


def golden_model():
    """Whole detector at a tiny configuration through the reference's own methods
    (models/model_3detr.py:1767-1794; the CLIP branch is gated off, weights absent):
    N=1024 points, 128 encoder tokens, 32 queries, enc 2 layers d=256, dec 3 layers d=64."""
    from golden.weights import fill_deterministic, grad_digest
    import models.model_3detr as M  # the REFERENCE module
    from datasets.sunrgbd_anonymous_aligned_image import SunrgbdAnonymousAlignedImageDatasetConfig

    args = _args()
    cfg = SunrgbdAnonymousAlignedImageDatasetConfig(if_print=False, args=args)

    # ---- kink margins.  A ReLU whose pre-activation lies within fp32 rounding of 0, or a max-pool
    # whose two largest DISTINCT candidates do, takes a different branch under any other summation
    # order (CPU vs GPU, conv vs GEMM), and in this 16k-row toy problem one such flip moves the
    # first-layer gradients by ~1 %.  The input scene is therefore chosen, among `SEARCH` seeds, as
    # the one whose smallest margin (in units of the tensor's std) is largest; the fixture then
    # holds at 1e-3 for any fp32 implementation.  Recorded by wrapping F.relu / F.max_pool2d BEFORE
    # the reference modules are constructed (the transformer layers bind F.relu at construction).
    import torch.nn.functional as F
    margins = []
    real_relu, real_pool = F.relu, F.max_pool2d

    def relu_probe(x, inplace=False):
        with torch.no_grad():
            margins.append(float(x.abs().min() / (x.std() + 1e-30)))
        return real_relu(x, inplace=inplace)

    def pool_probe(x, *a, **k):
        with torch.no_grad():
            top = x.topk(2, dim=-1).values if x.shape[-1] > 1 else None
            if top is not None:
                gap = top[..., 0] - top[..., 1]
                gap = gap[gap > 0]  # exact ties are padded duplicates of one row: identical everywhere
                if gap.numel():
                    margins.append(float(gap.min() / (x.std() + 1e-30)))
        return real_pool(x, *a, **k)

    F.relu, F.max_pool2d = relu_probe, pool_probe
    try:
        pre, enc, dec = M.build_preencoder(args), M.build_encoder(args), M.build_decoder(args)
        model = M.Model3DETRPredictedBoxDistillationHead(pre, enc, dec, cfg, encoder_dim=256,
                                                         decoder_dim=args.dec_dim, mlp_dropout=0.0,
                                                         num_queries=args.nqueries, if_with_clip_train=False,
                                                         args=args)
        fill_deterministic(model, seed=9)
        state0 = {k: v.clone() for k, v in model.state_dict().items()}

        def forward(inputs):
            point_clouds = inputs["point_clouds"]
            enc_xyz, enc_features, enc_inds = model.run_encoder(point_clouds)
            enc_features = model.encoder_to_decoder_projection(enc_features.permute(1, 2, 0)).permute(2, 0, 1)
            dims = [inputs["point_cloud_dims_min"], inputs["point_cloud_dims_max"]]
            query_xyz, query_embed = model.get_query_embeddings(enc_xyz, dims)
            enc_pos = model.pos_embedding(enc_xyz, input_range=dims).permute(2, 0, 1)
            query_embed = query_embed.permute(2, 0, 1)
            tgt = torch.zeros_like(query_embed)
            box_features = model.decoder(tgt, enc_features, query_pos=query_embed, pos=enc_pos)[0]
            pred = model.get_box_predictions(query_xyz, dims, box_features, point_clouds, inputs)
            return enc_xyz, enc_features, enc_inds, query_xyz, box_features, pred

        def scene(seed):
            pc, mn, mx = make_batch(2, 1024, seed=seed)
            return pc, mn, mx, {"point_clouds": torch.from_numpy(pc), "point_cloud_dims_min": torch.from_numpy(mn),
                                "point_cloud_dims_max": torch.from_numpy(mx)}

        SEARCH = int(os.environ.get("CODA_GOLDEN_SEARCH", "160"))
        best = (-1.0, None)
        for seed in range(31, 31 + SEARCH):
            model.load_state_dict(state0)
            worst = []
            for mode in ["train", "eval"]:  # the same two passes the fixture records
                model.train(mode == "train")
                margins.clear()
                with torch.no_grad():
                    forward(scene(seed)[3])
                worst.append(min(margins))
            if min(worst) > best[0]:
                best = (min(worst), seed)
        print(f"model_tiny: scene seed {best[1]} has the largest kink margin {best[0]:.2e} (of {SEARCH} seeds)")
        model.load_state_dict(state0)
    finally:
        F.relu, F.max_pool2d = real_relu, real_pool
    out = {"state_keys": np.array(sorted(model.state_dict().keys())),
           "state_shapes": np.array([str(tuple(model.state_dict()[k].shape)) for k in sorted(model.state_dict())]),
           "scene_seed": np.int32(best[1]), "kink_margin": np.float64(best[0])}
    pc, mn, mx, inputs = scene(best[1])
    for mode in ["train", "eval"]:
        model.train(mode == "train")
        model.zero_grad()
        enc_xyz, enc_features, enc_inds, query_xyz, box_features, pred = forward(inputs)
        o = pred["outputs"]
        out.update({f"{mode}_enc_xyz": _np(enc_xyz), f"{mode}_enc_inds": _np(enc_inds),
                    f"{mode}_enc_features": _np(enc_features), f"{mode}_query_xyz": _np(query_xyz),
                    f"{mode}_box_features": _np(box_features)})
        for k, v in o.items():
            if k != "point_clouds":
                out[f"{mode}_out/{k}"] = _np(v)
        for li, aux in enumerate(pred["aux_outputs"]):
            for k in ["sem_cls_logits", "center_normalized", "box_corners", "text_correlation_embedding"]:
                out[f"{mode}_aux{li}/{k}"] = _np(aux[k])
        if mode == "train":
            gen = torch.Generator().manual_seed(3)
            loss = 0
            for k in ["sem_cls_logits", "text_correlation_embedding", "center_normalized", "size_normalized",
                      "angle_logits", "angle_residual", "box_corners"]:
                w = torch.randn(o[k].shape, generator=gen)
                out[f"train_lossw/{k}"] = _np(w)
                loss = loss + (o[k] * w).sum()
                for aux in pred["aux_outputs"]:
                    loss = loss + 0.5 * (aux[k] * w).sum()
            loss.backward()
            out["train_loss"] = _np(loss)
            for k, v in grad_digest(model).items():
                out[f"train_grad/{k}"] = v
    # open-vocabulary scores (get_class_scores, :1742-1764) on synthetic unit-norm text embeddings
    gen = torch.Generator().manual_seed(4)
    text = torch.nn.functional.normalize(torch.randn(10, 512, generator=gen), dim=-1)
    pred["outputs"]["text_features_clip"] = text.unsqueeze(0).repeat(2, 1, 1)
    pred["outputs"]["logit_scale"] = torch.clip(torch.tensor(np.log(1 / 0.07)).exp(), max=100).float()
    _, scores, obj = model.get_class_scores(pred)
    out.update({"text_features": _np(text), "class_scores": _np(scores), "pc": pc, "dims_min": mn, "dims_max": mx})
    _save("model_tiny.npz", **out)


def golden_criterion():
    """The live loss terms of criterion.py through the reference's own methods, with the
    reference Hungarian Matcher (cost_class=1, cost_center=... as main.py defaults) on
    synthetic predictions / targets.  SetCriterion.__init__ hard-codes .to('cuda')
    (criterion.py:97), so the object is assembled with __new__ + attributes."""
    import criterion as RC  # the REFERENCE module
    from types import SimpleNamespace

    gen = torch.Generator().manual_seed(21)
    B, nq, ngt, ncls, nbin = 3, 24, 8, 10, 12
    outputs = {
        "sem_cls_logits": torch.randn(B, nq, 2, generator=gen).requires_grad_(True),
        "text_correlation_embedding": torch.randn(B, nq, 512, generator=gen).requires_grad_(True),
        "center_normalized": torch.rand(B, nq, 3, generator=gen).requires_grad_(True),
        "size_normalized": torch.rand(B, nq, 3, generator=gen).requires_grad_(True),
        "angle_logits": torch.randn(B, nq, nbin, generator=gen).requires_grad_(True),
        "angle_residual_normalized": torch.randn(B, nq, nbin, generator=gen).requires_grad_(True),
    }
    probs = torch.softmax(outputs["sem_cls_logits"].detach(), -1)
    outputs["sem_cls_prob"] = probs[..., :-1]
    outputs["objectness_prob"] = 1 - probs[..., -1]
    nactual = torch.tensor([5, 0, 8])
    present = (torch.arange(ngt)[None] < nactual[:, None]).float()
    targets = {
        "gt_box_present": present,
        "gt_box_sem_cls_label": torch.zeros(B, ngt, dtype=torch.int64),
        "gt_box_centers_normalized": torch.rand(B, ngt, 3, generator=gen),
        "gt_box_sizes_normalized": torch.rand(B, ngt, 3, generator=gen),
        "gt_angle_class_label": torch.randint(0, nbin, (B, ngt), generator=gen),
        "gt_angle_residual_label": (torch.rand(B, ngt, generator=gen) - 0.5) * 0.2,
        "gt_box_seen_sem_cls_label": torch.randint(0, ncls, (B, ngt), generator=gen),
        "gt_box_seen_sem_cls_confi": torch.rand(B, ngt, generator=gen),
        "text_features_clip": torch.nn.functional.normalize(torch.randn(ncls, 512, generator=gen), dim=-1)
        .unsqueeze(0).repeat(B, 1, 1),
        "logit_scale": torch.tensor(1 / 0.07).clamp(max=100),
        "gt_text_correlation_embedding": torch.nn.functional.normalize(torch.randn(B, nq, 512, generator=gen), dim=-1),
        "gt_text_correlation_embedding_mask": (torch.rand(B, nq, 1, generator=gen) < 0.25).float(),
        "weak_box_cate_label": torch.randint(0, ncls, (B, nq), generator=gen),
        "weak_confidence_weight": torch.rand(B, nq, generator=gen) * (torch.rand(B, nq, generator=gen) < 0.5),
    }
    targets["nactual_gt"] = present.sum(1).long()
    targets["num_boxes"] = float(max(int(targets["nactual_gt"].sum()), 1))
    targets["num_boxes_replica"] = int(targets["nactual_gt"].sum())
    crit = RC.SetCriterion.__new__(RC.SetCriterion)
    torch.nn.Module.__init__(crit)
    crit.dataset_config = SimpleNamespace(num_semcls=1, num_angle_bin=nbin)
    crit.register_buffer("semcls_percls_weights", torch.tensor([1.0, 0.25]))
    crit.confidence_type = "clip-max-prob"
    matcher = RC.Matcher(cost_class=1, cost_objectness=0, cost_giou=0, cost_center=1)
    outputs["gious"] = torch.zeros(B, nq, ngt)
    outputs["center_dist"] = torch.cdist(outputs["center_normalized"], targets["gt_box_centers_normalized"], p=1)
    assignments = matcher(outputs, targets)
    res = {}
    for name in ["loss_sem_cls_softmax_skip_none_gt_sample", "loss_angle", "loss_center", "loss_size",
                 "loss_cardinality", "loss_predicted_region_embed_l1",
                 "loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi"]:
        res.update(getattr(crit, name)(outputs, targets, assignments))
    total = sum(v for k, v in res.items() if k != "loss_cardinality")
    total.backward()
    out = {}
    for k, v in outputs.items():
        out[f"out/{k}"] = _np(v)
        if v.requires_grad and v.grad is not None:
            out[f"grad/{k}"] = _np(v.grad)
    for k, v in targets.items():
        out[f"tgt/{k}"] = _np(v) if isinstance(v, torch.Tensor) else np.asarray(v)
    out["assign/per_prop_gt_inds"] = _np(assignments["per_prop_gt_inds"])
    out["assign/proposal_matched_mask"] = _np(assignments["proposal_matched_mask"])
    for k, v in res.items():
        out[f"loss/{k}"] = _np(v)
    _save("criterion.npz", **out)


def golden_giou():
    """generalized_box3d_iou (utils/box_util.py:861-875 -> the TorchScript tensor path, :655-745) on
    seeded boxes built with the reference's own corner builder: rotated and axis-aligned, with
    overlapping pairs, padded GT columns and a degenerate box."""
    import utils.box_util as RB  # the REFERENCE module
    gen = torch.Generator().manual_seed(8)
    B, K1, K2 = 3, 12, 6
    out = {}

    def boxes(n, rotated):
        centre = torch.rand(B, n, 3, generator=gen) * 2.0
        size = torch.rand(B, n, 3, generator=gen) * 1.5 + 0.2
        angle = (torch.rand(B, n, generator=gen) - 0.5) * 3.0 if rotated else torch.zeros(B, n)
        return RB.get_3d_box_batch_tensor(size, angle, centre), angle

    for tag, rotated in [("rot", True), ("axis", False)]:
        c1, _ = boxes(K1, rotated)
        c2, ang2 = boxes(K2, rotated)
        c2[0, 1] = c1[0, 3]          # an identical pair (gIoU 1)
        c2[1, 0, 4:] = c2[1, 0, :4]  # zero-height GT box
        nums = torch.tensor([K2, 4, 0])
        g = RB.generalized_box3d_iou(c1, c2, nums, rotated_boxes=rotated, needs_grad=False)
        v = RB.generalized_box3d_iou(c1, c2, nums, rotated_boxes=rotated, return_inter_vols_only=True)
        out.update({f"{tag}_corners1": _np(c1), f"{tag}_corners2": _np(c2), f"{tag}_nums": _np(nums).astype(np.int32),
                    f"{tag}_gious": _np(g), f"{tag}_inter_vols": _np(v)})
    _save("giou.npz", **out)


EVAL_CONFIGS = {  # name -> overrides of utils/ap_calculator.py:1021-1051's defaults
    "default": {},                                   # 3-D NMS within a class, per-class proposals, empty boxes removed
    "nms3d": {"cls_nms": False},
    "nms2d": {"use_3d_nms": False},
    "old_type": {"use_old_type_nms": True, "nms_iou": 0.5},
    "keep_empty": {"remove_empty_box": False, "per_class_proposal": False},
    "no_nms": {"no_nms": True, "per_class_proposal": False, "use_cls_confidence_only": True},
}


def golden_eval_post():
    """parse_predictions / parse_predictions_obb (utils/ap_calculator.py:777-1018, :45-286) on a synthetic scene
    batch: proposals around the scene's furniture (some overlapping heavily, some in empty space, two of zero
    size), one scene whose proposals are all empty.  Stored: the inputs and, per configuration and scene, the
    (class, proposal index, score) rows of the returned lists in the returned order."""
    import utils.ap_calculator as RA  # the REFERENCE module (Delaunay in-hull test, utils/nms.py)
    import utils.box_util as RB
    gen = torch.Generator().manual_seed(17)
    B, K, N, ncls = 3, 40, 4096, 4
    pc, mn, mx = make_batch(B, N, seed=99)
    pts = torch.from_numpy(pc)
    # centres: half of them ON scene points (boxes that hold points), the rest anywhere in the scene's bbox
    pick = torch.randint(0, N, (B, K), generator=gen)
    on_points = torch.gather(pts, 1, pick.unsqueeze(-1).expand(-1, -1, 3))
    anywhere = torch.from_numpy(mn)[:, None] + torch.rand(B, K, 3, generator=gen) * torch.from_numpy(mx - mn)[:, None]
    centres = torch.where((torch.arange(K) % 2 == 0)[None, :, None], on_points, anywhere)
    centres[:, 1::8] = centres[:, 0::8][:, :centres[:, 1::8].shape[1]] + 0.05   # near-duplicates: NMS has work to do
    sizes = torch.rand(B, K, 3, generator=gen) * 1.2 + 0.3
    sizes[0, 5] = 0.0   # zero boxes (all corners at the origin: the only degenerate box both variants accept --
    sizes[1, 7] = 0.0   # qhull raises on any other flat hull)
    angles = (torch.rand(B, K, generator=gen) - 0.5) * 3.0
    centres[2] = torch.from_numpy(mx)[2] + 5.0 + torch.rand(K, 3, generator=gen)       # scene 2: nothing holds points
    centres[0, 5] = 0.0
    centres[1, 7] = 0.0
    cam = torch.stack((centres[..., 0], -centres[..., 2], centres[..., 1]), -1)        # depth -> upright camera
    corners = RB.get_3d_box_batch_tensor(sizes, angles, cam)
    probs = torch.softmax(torch.randn(B, K, ncls, generator=gen) * 2, -1)
    obj = torch.rand(B, K, generator=gen)
    obj[0, :6] = 0.01                                                                   # below conf_thresh
    out = {"corners": _np(corners), "points": pc, "sem_cls_probs": _np(probs), "objectness": _np(obj),
           "centers": _np(centres), "sizes": _np(sizes), "angles": _np(angles)}
    cfg_ds = types.SimpleNamespace(num_semcls=ncls)

    def rows(lists):
        res = []
        for i, lst in enumerate(lists):
            r = np.zeros((len(lst), 3), np.float64)
            for n, item in enumerate(lst):
                j = int(np.nonzero((_np(corners)[i] == item[1]).all(axis=(1, 2)))[0][0])
                r[n] = (item[0], j, item[2])
            res.append(r)
        return res

    for name, over in EVAL_CONFIGS.items():
        cfg = RA.get_ap_config_dict(dataset_config=cfg_ds, **over)
        for i, r in enumerate(rows(RA.parse_predictions(corners, probs, obj, pts, cfg))):
            out[f"{name}_plain_{i}"] = r
        cfg = RA.get_ap_config_dict(dataset_config=cfg_ds, **over)
        lists = RA.parse_predictions_obb(corners, probs, obj, pts, cfg, centres, sizes, angles)
        for i, r in enumerate(rows(lists)):
            out[f"{name}_obb_{i}"] = r
        out[f"{name}_obb_row0"] = _np(lists[0][0][3]) if lists[0] else np.zeros(0, np.float32)
    _save("eval_post.npz", **out)


def synthetic_camera(bsz, nbox, gen, image_size=(730, 531)):
    """Seeded stand-ins for the per-scene entries the dataset adds for the image branch
    (datasets/sunrgbd_anonymous_aligned_image.py:884-899): SUN RGB-D-like intrinsics / tilt, augmentation arrays."""
    def rnd(*shape):
        return torch.rand(*shape, generator=gen, dtype=torch.float64)
    K = torch.zeros(bsz, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = 520 + 20 * rnd(bsz)
    K[:, 1, 1] = 520 + 20 * rnd(bsz)
    K[:, 0, 2] = 320 + 10 * rnd(bsz)
    K[:, 1, 2] = 240 + 10 * rnd(bsz)
    K[:, 2, 2] = 1
    tilt = (rnd(bsz) - 0.5) * 0.3
    Rtilt = torch.zeros(bsz, 3, 3, dtype=torch.float64)
    Rtilt[:, 0, 0] = 1
    Rtilt[:, 1, 1] = torch.cos(tilt); Rtilt[:, 1, 2] = -torch.sin(tilt)
    Rtilt[:, 2, 1] = torch.sin(tilt); Rtilt[:, 2, 2] = torch.cos(tilt)
    ang = (rnd(bsz) - 0.5) * 0.6
    rot = torch.zeros(bsz, 3, 3, dtype=torch.float64)
    rot[:, 0, 0] = torch.cos(ang); rot[:, 0, 1] = -torch.sin(ang)
    rot[:, 1, 0] = torch.sin(ang); rot[:, 1, 1] = torch.cos(ang)
    rot[:, 2, 2] = 1
    flip = torch.where(rnd(bsz, 1) < 0.5, -1.0, 1.0).to(torch.float64)
    iflip = torch.where(flip < 0, 0.0, 1.0).to(torch.float64)
    ow = torch.full((bsz,), 640.0, dtype=torch.float64)
    oh = torch.full((bsz,), 480.0, dtype=torch.float64)
    return {"K": K, "Rtilt": Rtilt, "rot_array": rot, "scale_array": 0.9 + 0.2 * rnd(bsz, 1, 3), "flip_array": flip,
            "image_flip_array": iflip, "flip_length": torch.full((bsz,), float(image_size[0]), dtype=torch.float64),
            "ori_width": ow, "ori_height": oh, "y_offset": (image_size[0] - ow) // 2, "x_offset": (image_size[1] - oh) // 2}


def golden_clip_crops():
    """project_3dpoint_to_2dpoint_corners_tensor (datasets/sunrgbd_utils.py:611-635), the projection at the heart
    of the image branch, on seeded boxes in front of / beside / behind a seeded camera, after the un-augmentation of
    models/model_3detr.py:919-928 (replayed here with the reference's tensor expressions)."""
    import datasets.sunrgbd_utils as RU  # the REFERENCE module
    import utils.box_util as RB
    gen = torch.Generator().manual_seed(23)
    B, K = 3, 24
    cam = synthetic_camera(B, K, gen)
    centres = torch.stack(((torch.rand(B, K, generator=gen) - 0.5) * 6, 1.0 + torch.rand(B, K, generator=gen) * 5,
                           (torch.rand(B, K, generator=gen) - 0.5) * 2), -1)
    centres[:, :3, 1] = -2.0                                # behind the camera
    sizes = torch.rand(B, K, 3, generator=gen) * 1.5 + 0.2
    sizes[0, 5] = 0.0
    angles = (torch.rand(B, K, generator=gen) - 0.5) * 3
    corners = RB.get_3d_box_batch_tensor_xyz(sizes, angles, centres)        # box_corners_xyz of the model
    g = corners.detach().clone() * cam["scale_array"].unsqueeze(1)
    g = torch.matmul(g, cam["rot_array"].unsqueeze(1))
    g[:, :, :, 0] = g[:, :, :, 0] * cam["flip_array"].unsqueeze(-1)
    uv, depth = RU.project_3dpoint_to_2dpoint_corners_tensor(g.to(torch.double), K_tensor=cam["K"], Rtilt_tensor=cam["Rtilt"])
    out = {"corners": _np(corners), "sizes": _np(sizes), "uv_raw": _np(uv), "depth": _np(depth)}
    out.update({"cam_" + k: _np(v) for k, v in cam.items()})
    _save("clip_crops.npz", **out)


TOWER_CASES = {  # name: (resolution, patch, width, layers, heads, out_dim, images)
    "small": (64, 16, 128, 2, 2, 64, 3),          # 17 tokens
    "b16_tokens": (224, 16, 128, 1, 2, 32, 2),    # 197 tokens: the ragged length of ViT-B/16
    "patch32": (96, 32, 192, 2, 3, 48, 2),        # 10 tokens, three heads
}


def tower_images(name, n, res):
    return torch.randn(n, 3, res, res, generator=torch.Generator().manual_seed(zlib_seed(name)))


def zlib_seed(name):
    import zlib
    return zlib.crc32(name.encode()) & 0x7fffffff


def golden_clip_tower():
    """The reference's VisionTransformer (CLIP/clip/model.py:595-659) in float32 on seeded weights
    (tests/golden/weights.py) and seeded images: class-token embedding and all-token embeddings."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_clip_model", os.path.join(REF, "CLIP", "clip", "model.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from golden.weights import fill_deterministic
    out = {}
    for name, (res, patch, width, layers, heads, odim, n) in TOWER_CASES.items():
        vit = fill_deterministic(ref.VisionTransformer(res, patch, width, layers, heads, odim), seed=5).eval()
        with torch.no_grad():
            cls, tok = vit(tower_images(name, n, res))
        out[name + "_cls"], out[name + "_tokens"] = _np(cls), _np(tok)
        out[name + "_keys"] = np.array(["%s %s" % (k, tuple(v.shape)) for k, v in sorted(vit.state_dict().items())])
    _save("clip_tower.npz", **out)


def nms_restated(boxes, scores, iou_threshold):
    """torchvision.ops.nms (0.9.1, the version README.md:43 names; the package is not in this image): greedy
    suppression in descending score order, IoU = inter / (a1 + a2 - inter) on (x1,y1,x2,y2) float32 boxes without a
    +1, suppress when IoU > threshold; returns the kept indices in that order.  Equal scores: lower index first
    (torchvision's sort is not stable; the fixture inputs only tie on discarded boxes)."""
    boxes = boxes.detach().to(torch.float32)
    order = sorted(range(boxes.shape[0]), key=lambda j: (-float(scores[j]), j))
    x1, y1, x2, y2 = boxes.unbind(1)
    areas = (x2 - x1) * (y2 - y1)
    dead, keep = set(), []
    for a, i in enumerate(order):
        if i in dead:
            continue
        keep.append(i)
        for j in order[a + 1:]:
            if j in dead:
                continue
            w = torch.clamp(torch.minimum(x2[i], x2[j]) - torch.maximum(x1[i], x1[j]), min=0)
            h = torch.clamp(torch.minimum(y2[i], y2[j]) - torch.maximum(y1[i], y1[j]), min=0)
            inter = w * h
            if float(inter / (areas[i] + areas[j] - inter)) > iou_threshold:
                dead.add(j)
    return torch.tensor(keep, dtype=torch.int64)


def golden_region_branch():
    """The reference's two image-branch methods (models/model_3detr.py:902-1210 and :1212-1632) run END TO END on
    seeded inputs (tests/golden/region_inputs.py) with a seeded stand-in image tower: selection (random /
    objectness-driven), crops, embedding scatter + mask, novel-box discovery into the ground truth, CLIP weak labels,
    and the stage-2 pseudo-label mining (2-D NMS, 3-D IoU against the ground truth, objectness / CLIP thresholds, the
    .npy rows).  Stand-ins for what this image lacks: torchvision's ``Resize`` / ``nms`` and CLIP's tensor transform
    (restated: oracle/crop_oracle.py, nms_restated above)."""
    import tempfile
    import types as T
    import models.model_3detr as M
    import datasets.sunrgbd_utils as RU
    import utils.box_util as RB
    from golden import region_inputs as R
    from oracle import crop_oracle as CO
    cls = M.Model3DETRPredictedBoxDistillationHead
    sys.modules["torchvision.ops"].nms = nms_restated
    sys.modules["torchvision"].ops = sys.modules["torchvision.ops"]

    def resize(img):  # torchvision 0.9.1 Resize(224, BICUBIC) on a square uint8 (3,E,E) tensor
        y = torch.nn.functional.interpolate(img.unsqueeze(0).float(), size=(224, 224), mode="bicubic", align_corners=False)
        return y.clamp(0, 255).round()[0].to(torch.uint8)

    def preprocess(x):  # CLIP/clip/clip.py:95-101 on (n,3,224,224): resize / crop are identities at this size
        x = x / 255.0
        return (x - torch.tensor(CO.MEAN).view(1, 3, 1, 1)) / torch.tensor(CO.STD).view(1, 3, 1, 1)

    out = {}
    for name, (method, epoch, flags) in R.CASES.items():
        inputs, outputs, tower_w = R.build(RB.get_3d_box_batch_tensor_xyz, RB.get_3d_box_batch_tensor)
        text = outputs.pop("text_features_all")
        ncls = R.NTEXT if (name.startswith("stage1_objectness") or name.startswith("stage2")) else R.NSEEN
        outputs["text_features_clip"] = text[:ncls].unsqueeze(0).repeat(R.B, 1, 1)   # (stage 2: the superset, :1800-1802)
        outputs["maybe_novel_text_features_clip"] = text
        me = T.SimpleNamespace(device="cpu", dataset_util=RU, box_idx_list=np.arange(128, dtype=np.int8),
                               resize=resize, preprocess_for_tensor=preprocess, clip_model=R.StandInTower(tower_w),
                               if_select_box_by_objectness=False, if_keep_box=False, if_clip_weak_labels=False,
                               if_accumulate_former_pseudo_labels=False, **R.MODEL_FLAGS)
        for k, v in flags.items():
            setattr(me, k, v)
        me.cal_iou = T.MethodType(cls.cal_iou, me)
        tmp = tempfile.mkdtemp()
        inputs["pseudo_box_path"] = [os.path.join(tmp, f"scene{b}.npy") for b in range(R.B)]
        if flags.get("if_accumulate_former_pseudo_labels"):
            np.save(inputs["pseudo_box_path"][0], np.zeros((0, 10)))
            np.save(inputs["pseudo_box_path"][1], np.zeros((0, 10)))
            np.save(inputs["pseudo_box_path"][2], np.arange(10, dtype=np.float64)[None])
        np.random.seed(2024)
        with torch.no_grad():
            res = getattr(cls, method)(me, inputs, outputs, curr_epoch=epoch)
        out[f"{name}/emb"] = _np(res["gt_text_correlation_embedding"])
        out[f"{name}/mask"] = _np(res["gt_text_correlation_embedding_mask"])
        if "weak_box_cate_label" in res:  # (the stage-2 method adds no weak entries without if_clip_weak_labels, :1614)
            out[f"{name}/weak_label"] = _np(res["weak_box_cate_label"])
            out[f"{name}/weak_conf"] = _np(res["weak_confidence_weight"])
        for k in R.GT_KEYS:
            out[f"{name}/{k}"] = _np(inputs[k])
        for b, path in enumerate(inputs["pseudo_box_path"]):
            out[f"{name}/pseudo{b}"] = np.load(path) if os.path.exists(path) else np.zeros((0, 10))
        print(name, "masked", int(res["gt_text_correlation_embedding_mask"].sum()), "gt now",
              [int(x) for x in inputs["gt_box_present"].sum(1)], "pseudo rows",
              [out[f"{name}/pseudo{b}"].shape[0] for b in range(R.B)], "weak conf>0", int((res["weak_confidence_weight"] > 0).sum()) if "weak_confidence_weight" in res else None)
    _save("region_branch.npz", **out)


def golden_eval_det():
    """The reference's APCalculator.accumulate / compute_metrics / metrics_to_str (utils/ap_calculator.py:1491-1803)
    over utils/eval_det.eval_det with its own box3d_iou (numpy clip + scipy ConvexHull), on the seeded detection lists
    of tests/golden/eval_inputs.py: every metric of both IoU thresholds, and the printed table."""
    import types as T
    from utils.ap_calculator import APCalculator
    from golden import eval_inputs as E
    cfg = T.SimpleNamespace(num_semcls=E.NCLS)
    calc = APCalculator(dataset_config=cfg, ap_iou_thresh=[0.25, 0.5], class2type_map=None, exact_eval=False,
                        args=T.SimpleNamespace(dataset_name="sunrgbd"))
    for pred, gt in E.build():
        calc.accumulate(pred, gt)
    ret = calc.compute_metrics()
    out = {}
    for t, d in ret.items():
        out[f"keys_{t}"] = np.array(list(d.keys()))
        out[f"vals_{t}"] = np.array([float(v) for v in d.values()], dtype=np.float64)
    out["table"] = np.array(calc.metrics_to_str(ret))
    out["dict"] = np.array([calc.metrics_to_dict(ret)[k] for k in sorted(calc.metrics_to_dict(ret))])
    print("mAP", {t: float(d["mAP"]) for t, d in ret.items()}, "classes", sum(k.endswith("Average Precision") for k in ret[0.25]))
    _save("eval_det.npz", **out)


def golden_step_full(case="configs2", f64=False):
    """ONE WHOLE TRAINING STEP AT BASELINE.json configs[2]'s / configs[3]'s PER-GPU SIZE THROUGH THE REFERENCE'S OWN MODULES (8 scenes x
    20 000 points, 2048 encoder tokens, 256 queries, 3 + 8 layers, the stage-2 loss set): models/model_3detr.py's
    pre-encoder / encoder / decoder / heads (:1767-1794, the oracle's C ops behind pointnet2._ext), criterion.py's
    SetCriterion.forward (:1162-1216: gIoU, Hungarian matching of all 8 layers, matched box terms, both CLIP-space
    alignment terms) and backward, float32 on the CPU.  The CLIP image branch (checkpoint absent) is replaced by the
    same seeded tensors tests/test_full_step_gpu.py hands to the product's region-embedding seam.  Inputs are NOT
    stored: tests/golden/step_inputs.py rebuilds them from seeds.  Stored: loss, every loss term, the 64 assignments,
    the pre-encoder's FPS indices, output digests and per-parameter gradient digests (sum, norm, 1024 samples)."""
    import bench
    import criterion as RC  # the REFERENCE module
    import models.model_3detr as M  # the REFERENCE module
    from datasets.sunrgbd_anonymous_aligned_image import SunrgbdAnonymousAlignedImageDatasetConfig
    from golden import step_inputs as SI
    from golden.weights import fill_deterministic

    args = SI.recipe(case)
    for k, v in vars(_args()).items():  # flags the reference reads beyond the hot path's
        if not hasattr(args, k):
            setattr(args, k, v)
    for k, v in dict(only_image_class=False, only_prompt_loss=False, if_skip_no_seen_scene_objectness=False,
                     if_only_seen_in_loss=False).items():
        if not hasattr(args, k):
            setattr(args, k, v)
    cfg = SunrgbdAnonymousAlignedImageDatasetConfig(if_print=False, args=args)
    pre, enc, dec = M.build_preencoder(args), M.build_encoder(args), M.build_decoder(args)
    model = M.Model3DETRPredictedBoxDistillationHead(pre, enc, dec, cfg, encoder_dim=args.enc_dim,
                                                     decoder_dim=args.dec_dim, mlp_dropout=0.0,
                                                     num_queries=args.nqueries, if_with_clip_train=False, args=args)
    fill_deterministic(model, seed=SI.WEIGHT_SEED)
    batch, seam = SI.build(case)
    # the test point is moved off the ReLU kinks of the query projection (SI.condition_query_projection: why); the two
    # conditioned biases travel in the fixture
    qp_bias = SI.condition_query_projection(model, batch, SI.CASES[case]["nq"])
    if f64:  # the float32 fixture's conditioned biases, so that both fixtures describe the same test point
        z32 = np.load(os.path.join(HERE, f"step_full_{case}.npz"))
        lin = [m for m in model.query_projection.layers if isinstance(m, (torch.nn.Conv1d, torch.nn.Linear))]
        with torch.no_grad():
            lin[0].bias.copy_(torch.from_numpy(z32["cond/qp_bias0"]))
            lin[1].bias.copy_(torch.from_numpy(z32["cond/qp_bias2"]))
        model.double()
        batch = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
        seam = {k: (v.double() if v.is_floating_point() else v) for k, v in seam.items()}
    model.train()
    # SetCriterion.__init__ parks a scratch tensor on 'cuda' (criterion.py:97): keep it on the host while constructing
    real_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: self if (a and a[0] == "cuda") else real_to(self, *a, **k)
    try:
        crit = RC.build_criterion(args, cfg)
    finally:
        torch.Tensor.to = real_to
    if f64:
        crit.double()  # (class weights and other float buffers of the criterion)
        # criterion.py:604 casts the text embeddings `.to(torch.float32)` in front of a bmm (lossless here: the seam's
        # embeddings ARE float32 data): let that one product promote instead of raising
        real_bmm = torch.bmm
        torch.bmm = lambda a, b, **kw: real_bmm(a.double(), b.double(), **kw)
    captured = {}
    real_match = crit.matcher.forward

    def spy(outputs, targets):
        res = real_match(outputs, targets)
        captured.setdefault("inds", []).append(res["per_prop_gt_inds"].clone())
        captured.setdefault("mask", []).append(res["proposal_matched_mask"].clone())
        return res

    crit.matcher.forward = spy
    import time
    t0 = time.time()
    point_clouds = batch["point_clouds"]
    enc_xyz, enc_features, enc_inds = model.run_encoder(point_clouds)
    enc_features = model.encoder_to_decoder_projection(enc_features.permute(1, 2, 0)).permute(2, 0, 1)
    dims = [batch["point_cloud_dims_min"], batch["point_cloud_dims_max"]]
    query_xyz, query_embed = model.get_query_embeddings(enc_xyz, dims)
    enc_pos = model.pos_embedding(enc_xyz, input_range=dims).permute(2, 0, 1)
    query_embed = query_embed.permute(2, 0, 1)
    box_features = model.decoder(torch.zeros_like(query_embed), enc_features, query_pos=query_embed, pos=enc_pos)[0]
    pred = model.get_box_predictions(query_xyz, dims, box_features, point_clouds, batch)
    o = pred["outputs"]
    o["logit_scale"] = torch.tensor(SI.LOGIT_SCALE)  # models/model_3detr.py:1796 with a released checkpoint's ln(100)
    o["text_features_clip"] = seam["text"][:args.train_range_max].unsqueeze(0).repeat(SI.B, 1, 1)  # :1803
    o["gt_text_correlation_embedding"] = seam["img_emb"]          # what get_predicted_box_clip_embedding adds (:1816)
    o["gt_text_correlation_embedding_mask"] = seam["mask"]
    o["weak_box_cate_label"] = seam["weak_label"]
    o["weak_confidence_weight"] = seam["weak_conf"]
    print(f"step_full: forward {time.time() - t0:.1f} s")
    loss, loss_dict = crit(pred, batch)
    print(f"step_full: criterion {time.time() - t0:.1f} s, loss {float(loss):.6f}")
    loss.backward()
    print(f"step_full: backward {time.time() - t0:.1f} s")
    if f64:
        out = {"loss": np.float64(float(loss)), "sa_inds": _np(enc_inds).astype(np.int32)}
        for name, p in model.named_parameters():
            if p.grad is not None:
                g = p.grad.detach().double().reshape(-1).numpy()
                out[f"norm/{name}"] = np.float64(np.linalg.norm(g))
                out[f"sketch/{name}"] = sketch(g, name)
        _save(f"step_full_{case}_f64.npz", **out)
        return
    out = {"loss": np.float64(float(loss)), "sa_inds": _np(enc_inds).astype(np.int32),
           # SetCriterion.forward matches the last layer first, then aux 0..6 (:1200-1210): stored in LAYER order
           "assign_inds": _np(torch.cat(captured["inds"][1:] + captured["inds"][:1])).astype(np.int16),
           "assign_mask": _np(torch.cat(captured["mask"][1:] + captured["mask"][:1])).astype(np.uint8),
           "cond/qp_bias0": _np(qp_bias[0]), "cond/qp_bias2": _np(qp_bias[1]),
           "loss_keys": np.array(sorted(loss_dict)),
           "loss_vals": np.array([float(loss_dict[k]) for k in sorted(loss_dict)], dtype=np.float64)}
    for k in ["sem_cls_logits", "text_correlation_embedding", "center_normalized", "size_normalized", "angle_logits",
              "angle_residual", "box_corners"]:
        t = o[k].detach().double()
        idx = np.linspace(0, t.numel() - 1, SI.SAMPLES).astype(np.int64)
        out[f"out/{k}"] = np.concatenate([[float(t.sum()), float(t.norm())], t.reshape(-1)[idx].numpy()])
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().double().reshape(-1)
        idx = np.linspace(0, g.numel() - 1, min(SI.SAMPLES, g.numel())).astype(np.int64)
        out[f"grad/{name}"] = np.concatenate([[float(g.sum()), float(g.norm())], g[idx].numpy()]).astype(np.float32 if g.numel() > 64 else np.float64)
    _save(f"step_full_{case}.npz", **out)


def sketch(g, name, k=128):
    """Count sketch of a flat float64 tensor: every element is added, with a seeded random sign, to one of k seeded
    random buckets.  For two tensors a, b:  sum_b (sketch(a) - sketch(b))_b^2  is an unbiased estimate of |a - b|^2
    (relative standard deviation ~ sqrt(2 / k) = 12.5 % on the square, 6 % on the norm) -- EVERY element takes part, in
    1 KB per tensor and O(n) work."""
    import zlib
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    rng = np.random.default_rng(zlib.crc32(name.encode()) + 77)
    bucket = rng.integers(0, k, g.size)
    sign = rng.integers(0, 2, g.size).astype(np.float64) * 2.0 - 1.0
    return np.bincount(bucket, weights=sign * g, minlength=k)


def golden_step_full_f64(case="configs2"):
    """the whole step of golden_step_full through the reference's modules IN FLOAT64 (run with
    `make_golden.py step_full_f64`, which installs the dtype-following operators; 10-20 minutes and ~20 GB on 8 cores),
    so that tests/test_reference_step_gpu.py can hold the product's gradients against the reference's TRUE gradients
    on WHOLE tensors at the north-star 1e-3: per parameter gradient its norm and a 128-bucket count sketch (see `sketch`)."""
    golden_step_full(case, f64=True)


def golden_step_full_configs3_f64():
    golden_step_full("configs3", f64=True)


def golden_step_full_configs4_f64():
    golden_step_full("configs4_fp32", f64=True)


def golden_step_full_configs3():
    golden_step_full("configs3")


def golden_step_full_configs4():
    golden_step_full("configs4_fp32")


if __name__ == "__main__":
    O.build()
    O.set_fma_mode(FMA_MODE)
    install_reference(any_dtype=any(w.endswith("_f64") for w in sys.argv[1:]))
    torch.set_num_threads(8)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    which = sys.argv[1:] or ["ops", "sa_module", "sa_module_wide", "transformer", "model", "criterion", "giou", "eval_post", "clip_crops", "clip_tower", "region_branch", "eval_det", "step_full", "step_full_configs3", "step_full_configs4"]
    for w in which:
        globals()["golden_" + w]()
