"""CoDA / 3DETR detector: the forward hot path.

Mirror of models/model_3detr.py for the model the CoDA scripts build
(``Model3DETRPredictedBoxDistillationHead``, :130-1833) restricted to the hot
path of SURVEY.md section 8: ``run_encoder`` (:535-555), the encoder->decoder
projection (:409-419, 1770-1772), ``get_query_embeddings`` (:513-526), the
decoder call (:1784-1792), ``get_box_predictions`` (:1634-1740),
``get_class_scores`` (:1742-1764), ``BoxProcessor`` (:56-127) and the builders
(:3935-4074).  Class names, constructor keywords, the ``forward`` signature, the
output dictionary keys and every ``state_dict`` key of the trunk are the
reference's, so ``main.py`` / ``engine.py`` can call it unchanged.

The CLIP side enters through two seams: ``text_features_fg_norm`` (and
``superset_text_features_fg_norm``: the normalised text embeddings, computed once
at init in the reference, :339-360 -- the text tower runs once at start-up and is
the deployment's) and ``region_embedding_provider``, the image-crop distillation
branch (``get_predicted_box_clip_embedding*``, :902-1632): ``clip_crops.
RegionEmbeddingProvider`` implements both of the reference's methods on the
device; any callable with its signature can stand in (tests, ``bench.py``).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib, box_decode, fused_bn_mlp
from .helpers import GenericMLP
from .pointnet2.pointnet2_modules import PointnetSAModuleVotes
from .pointnet2.pointnet2_utils import SamplingPrefetcher, furthest_point_sample
from .position_embedding import PositionEmbeddingCoordsSine, scale_points, shift_scale_points
from .transformer import (MaskedTransformerEncoder, TransformerDecoder, TransformerDecoderLayer,
                          TransformerEncoder, TransformerEncoderLayer)


class BoxProcessor(object):
    """Turns the heads' raw outputs into boxes (the torch formulation; ``box_decode.py`` is the fused one).
    Method names and signatures are the reference's (models/model_3detr.py:56-127): ``main.py`` / the evaluation
    code receive this object from ``build_model``."""

    def __init__(self, dataset_config):
        self.dataset_config = dataset_config

    def compute_predicted_center(self, center_offset, query_xyz, point_cloud_dims):
        """offsets are relative to the query position; also returned in units of the scene's bounding box"""
        absolute = query_xyz + center_offset
        return shift_scale_points(absolute, src_range=point_cloud_dims), absolute

    def compute_predicted_size(self, size_normalized, point_cloud_dims):
        extent = (point_cloud_dims[1] - point_cloud_dims[0]).clamp(min=1e-1)
        return scale_points(size_normalized, mult_factor=extent)

    def compute_predicted_angle(self, angle_logits, angle_residual):
        nbin = angle_logits.shape[-1]
        if nbin == 1:  # datasets without rotation: angle 0, with both heads kept in the autograd graph
            return (angle_logits * 0 + angle_residual * 0).squeeze(-1).clamp(min=0)
        winner = angle_logits.argmax(dim=-1).detach()
        angle = winner * (2 * np.pi / self.dataset_config.num_angle_bin) \
            + angle_residual.gather(2, winner.unsqueeze(-1)).squeeze(-1)
        # wrap into (-pi, pi] without boolean indexing (which would synchronise with the host)
        return torch.where(angle > np.pi, angle - 2 * np.pi, angle)

    def compute_objectness_and_cls_prob(self, cls_logits):
        prob = torch.softmax(cls_logits, dim=-1)
        return prob[..., :-1], 1 - prob[..., -1]  # (class probabilities, 1 - P(background))

    def box_parametrization_to_corners(self, box_center_unnorm, box_size_unnorm, box_angle):
        return self.dataset_config.box_parametrization_to_corners(box_center_unnorm, box_size_unnorm, box_angle)

    def box_parametrization_to_corners_xyz(self, box_center_unnorm, box_size_unnorm, box_angle):
        return self.dataset_config.box_parametrization_to_corners_xyz(box_center_unnorm, box_size_unnorm, box_angle)


class Model3DETRPredictedBoxDistillationHead(nn.Module):
    """pre_encoder (set abstraction) -> encoder -> projection -> query embeddings ->
    decoder -> MLP heads (box parameters + 512-d CLIP-space region embedding)."""

    def __init__(self, pre_encoder, encoder, decoder, dataset_config, image_text_encoder=None,
                 encoder_dim=256, decoder_dim=256, position_embedding="fourier", mlp_dropout=0.3,
                 num_queries=256, if_with_clip=False, if_use_gt_box=False, if_expand_box=False,
                 if_with_clip_embed=False, if_with_clip_train=True, num_cls_predict=1,
                 if_with_fake_classes=False, pooling_methods="average", if_clip_more_prompts=False,
                 if_keep_box=False, if_select_box_by_objectness=False, keep_objectness=0.5,
                 online_nms_update_novel_label=False, online_nms_update_accumulate_novel_label=False,
                 online_nms_update_accumulate_epoch=10, distillation_box_num=32, args=None,
                 text_features_fg_norm=None, region_embedding_provider=None, clip_model=None, logit_scale=None,
                 superset_text_features_fg_norm=None):
        super().__init__()
        self.if_with_fake_classes = if_with_fake_classes
        self.num_cls_predict = num_cls_predict
        self.pre_encoder = pre_encoder
        self.encoder = encoder
        self.args = args
        self.if_with_clip = if_with_clip
        self.if_with_clip_train = if_with_clip_train
        self.if_keep_box = if_keep_box  # poked by main.py:356
        self.if_select_box_by_objectness = if_select_box_by_objectness
        self.keep_objectness = keep_objectness
        self.distillation_box_num = distillation_box_num
        self.eval_layer_id = getattr(args, "eval_layer_id", -1) if args is not None else -1
        # --if_clip_superset (models/model_3detr.py:282-360): the training prompts are a larger class list (LVIS-derived,
        # 232 / 1201 entries) whose normalised embeddings the caller hands in like `text_features_fg_norm`
        self.if_clip_superset = bool(getattr(args, "if_clip_superset", False)) and superset_text_features_fg_norm is not None
        self.online_nms_update_save_novel_label_clip_driven_with_cate_confidence = bool(
            getattr(args, "online_nms_update_save_novel_label_clip_driven_with_cate_confidence", False))  # :437

        # The CLIP towers themselves are outside the hot path (weights are not redistributable here):
        # the constructor takes their PRODUCTS -- the normalised prompt embeddings, the frozen
        # temperature -- and, optionally, the loaded CLIP module, which is then exposed under the
        # attribute names main.py / engine.py poke (`clip_model`, `res_encoder`, engine.py:85-117;
        # absent attributes take the reference's own "no clip here" branch).  `CLIP_LOADER` below lets
        # build_model(args, dataset_config) obtain them without extra keyword arguments.
        self.region_embedding_provider = region_embedding_provider
        if clip_model is not None:
            for prm in clip_model.parameters():
                prm.requires_grad = False           # models/model_3detr.py:330-331
            self.clip_model = clip_model
            self.res_encoder = clip_model.visual    # :333
            if logit_scale is None:
                logit_scale = clip_model.logit_scale  # aliased, as in the reference (:367)
        if text_features_fg_norm is not None:
            self.register_buffer("text_features_fg_norm", text_features_fg_norm.to(torch.float32),
                                 persistent=False)
            self.train_range_max = int(getattr(args, "train_range_max", 0) or text_features_fg_norm.shape[0])
            self.test_range_max = text_features_fg_norm.shape[0]
            if isinstance(logit_scale, nn.Parameter):
                self.logit_scale = logit_scale
            else:
                # released CLIP checkpoints carry logit_scale = ln(100) (temperature 100 after the clip at
                # :1796); a caller with another checkpoint passes its value
                value = math.log(100.0) if logit_scale is None else float(logit_scale)
                self.logit_scale = nn.Parameter(torch.ones([]) * value, requires_grad=False)
        else:
            self.text_features_fg_norm = None
        if superset_text_features_fg_norm is not None:
            self.register_buffer("superset_text_features_fg_norm", superset_text_features_fg_norm.to(torch.float32),
                                 persistent=False)

        self.encoder_to_decoder_projection = GenericMLP(
            input_dim=256, hidden_dims=[512, 512], output_dim=decoder_dim, norm_fn_name="bn1d",
            activation="relu", use_conv=True, output_use_activation=True, output_use_norm=True,
            output_use_bias=False)
        self.pos_embedding = PositionEmbeddingCoordsSine(d_pos=decoder_dim, pos_type=position_embedding,
                                                         normalize=True)
        self.query_projection = GenericMLP(input_dim=decoder_dim, hidden_dims=[decoder_dim],
                                           output_dim=decoder_dim, use_conv=True,
                                           output_use_activation=True, hidden_use_bias=True)
        self.decoder = decoder
        self.build_mlp_heads(dataset_config, decoder_dim, mlp_dropout)
        self.num_queries = num_queries
        self.box_processor = BoxProcessor(dataset_config)

    def build_mlp_heads(self, dataset_config, decoder_dim, mlp_dropout):
        """Six heads on the decoder features, each Conv1d(k=1)+BN+ReLU+Dropout x2 then a linear output: objectness /
        class logits (+1 = "not an object"), centre offset, size, angle bin, per-bin angle residual, and the 512-d
        embedding in CLIP's joint space.  The registration order fixes the ``state_dict`` layout."""
        if self.if_with_fake_classes:
            self.num_cls_predict += 1
        widths = {"sem_cls_head": self.num_cls_predict + 1, "center_head": 3, "size_head": 3,
                  "angle_cls_head": dataset_config.num_angle_bin, "angle_residual_head": dataset_config.num_angle_bin,
                  "text_correlation_head": 512}
        self.mlp_heads = nn.ModuleDict(
            (name, GenericMLP(input_dim=decoder_dim, hidden_dims=[decoder_dim, decoder_dim], output_dim=width,
                              norm_fn_name="bn1d", activation="relu", use_conv=True, dropout=mlp_dropout))
            for name, width in widths.items())

    def get_query_embeddings(self, encoder_xyz, point_cloud_dims):
        ahead = getattr(self, "_queries_ahead", None)
        self._queries_ahead = None
        if ahead is not None and ahead[0] is encoder_xyz and ahead[1].shape[1] == self.num_queries:
            if len(ahead) > 2:  # sampled on the side stream during this very forward (run_pre_encoder)
                cur = torch.cuda.current_stream(encoder_xyz.device)
                cur.wait_event(ahead[2])
                ahead[1].record_stream(cur)
            query_inds = ahead[1].long()  # sampled next to the pre-encoder's own sampling (prefetch_sampling)
        else:
            query_inds = furthest_point_sample(encoder_xyz, self.num_queries).long()
        query_xyz = torch.gather(encoder_xyz, 1, query_inds.unsqueeze(-1).expand(-1, -1, 3))
        pos_embed = self.pos_embedding(query_xyz, input_range=point_cloud_dims)
        if self.query_projection.tokens_supported():
            # (B, C, nq) -> query-major tokens; the caller's permute(2, 0, 1) undoes the view below
            query_embed = self.query_projection.forward_tokens(pos_embed.permute(2, 0, 1)).permute(1, 2, 0)
        else:
            query_embed = self.query_projection(pos_embed)
        return query_xyz, query_embed

    def _break_up_pc(self, pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    def prefetch_sampling(self, inputs, wait_for="current"):
        """Optional: run the pre-encoder's sampling / grouping of an upcoming batch on a side
        stream (pointnet2_utils.SamplingPrefetcher); ``forward`` on the same
        ``inputs["point_clouds"]`` tensor then finds it ready.  Purely a scheduling aid: outputs
        are identical with and without it (up to fp32 summation order in the batch statistics)."""
        pc = inputs["point_clouds"]
        if not pc.is_cuda or pc.size(-1) > 3 or not hasattr(self.pre_encoder, "prepare"):
            return
        if not hasattr(self, "_sampling_prefetcher"):
            self._sampling_prefetcher = SamplingPrefetcher()
        after = None
        if type(self.encoder).__name__ == "TransformerEncoder" and os.environ.get("CODA_PREFETCH_QUERIES", "1") != "0":
            # an encoder that hands its xyz through unchanged: the object queries are a sampling of the
            # pre-encoder's centres, known before the encoder runs -- 140 us of one-workgroup-per-scene work that
            # would otherwise sit in line between encoder and decoder
            def after(prepared):
                prepared["query_inds"] = furthest_point_sample(prepared["new_xyz"], self.num_queries)
        self._sampling_prefetcher.submit(pc, self.pre_encoder, wait_for, after=after)

    def run_pre_encoder(self, point_clouds):
        """The set-abstraction stage alone: -> (xyz (B,M,3), features (B,C,M), inds (B,M)).  Its result
        can be handed back to ``forward(..., pre_encoded=...)``: the stage has data-dependent shapes
        (de-duplicated groups) while everything behind it is static."""
        xyz, features = self._break_up_pc(point_clouds)
        prepared = None
        if hasattr(self, "_sampling_prefetcher"):
            prepared = self._sampling_prefetcher.take(point_clouds)
        if prepared is not None:
            if "query_inds" in prepared:
                self._queries_ahead = (prepared["new_xyz"], prepared["query_inds"])
            return self.pre_encoder(prepared["xyz"], features, prepared=prepared)
        if (features is None and xyz.is_cuda and hasattr(self.pre_encoder, "prepare")
                and type(self.encoder).__name__ == "TransformerEncoder"
                and os.environ.get("CODA_PREFETCH_QUERIES", "1") != "0"):
            # No caller-side prefetch (an unchanged engine.py): the front runs in line, and the object queries -- a
            # sampling of the pre-encoder's CENTRES, 0.15-0.24 ms of one-workgroup-per-scene work -- go to the sampling
            # side stream as soon as the centres exist, under the shared MLP and the encoder instead of between encoder
            # and decoder.  Same kernels, same values.
            with torch.no_grad():  # (parameter-free; the cloud itself carries no gradient)
                front = self.pre_encoder.prepare(xyz) if not xyz.requires_grad else None
            if front is not None:
                main = torch.cuda.current_stream(xyz.device)
                side = getattr(self, "_query_stream", None)
                if side is None or side.device != xyz.device:
                    side = self._query_stream = torch.cuda.Stream(
                        device=xyz.device, priority=int(os.environ.get("CODA_PREFETCH_PRIORITY", "-1")))
                side.wait_stream(main)
                with torch.cuda.stream(side), torch.no_grad():
                    q_inds = furthest_point_sample(front["new_xyz"], self.num_queries)
                    done = torch.cuda.Event()
                    done.record(side)
                front["new_xyz"].record_stream(side)
                self._queries_ahead = (front["new_xyz"], q_inds, done)
                return self.pre_encoder(front["xyz"], features, prepared=front)
        return self.pre_encoder(xyz, features)

    def run_encoder(self, point_clouds, pre_encoded=None):
        if pre_encoded is None:
            pre_encoded = self.run_pre_encoder(point_clouds)
        pre_enc_xyz, pre_enc_features, pre_enc_inds = pre_encoded
        # (B, C, npoints) -> (npoints, B, C) for the seq-first transformer
        pre_enc_features = pre_enc_features.permute(2, 0, 1)
        enc_xyz, enc_features, enc_inds = self.encoder(pre_enc_features, xyz=pre_enc_xyz)
        if enc_inds is None:
            enc_inds = pre_enc_inds  # no down-sampling inside the encoder
        else:
            enc_inds = torch.gather(pre_enc_inds, 1, enc_inds.long())
        return enc_xyz, enc_features, enc_inds

    def _project_encoder_features(self, enc_features):
        """(npoints, B, C) -> (npoints, B, dec_dim) through encoder_to_decoder_projection
        (models/model_3detr.py:1866-1868)."""
        parsed = fused_bn_mlp.eligible([self.encoder_to_decoder_projection], enc_features)
        if parsed is not None and parsed[0][1] is None:
            npoints, batch, channel = enc_features.shape
            out = fused_bn_mlp.hidden_stack(enc_features.reshape(-1, channel), parsed)
            return out.view(npoints, batch, -1)
        return self.encoder_to_decoder_projection(enc_features.permute(1, 2, 0)).permute(2, 0, 1)

    def get_box_predictions(self, query_xyz, point_cloud_dims, box_features, point_clouds=None,
                            inputs=None):
        """box_features: (num_layers, num_queries, batch, channel) -> output dicts."""
        num_layers, num_queries, batch, channel = box_features.shape
        heads = self.mlp_heads
        # the narrow box heads first (their final layers run as one batched GEMM), the 512-d head last
        names = ["sem_cls_head", "center_head", "size_head", "angle_cls_head", "angle_residual_head",
                 "text_correlation_head"]
        parsed = fused_bn_mlp.eligible([heads[n] for n in names], box_features)
        if parsed is not None and all(tail is not None for _, tail in parsed):
            # all six heads at once on the decoder's own (layer, query, scene) token order:
            # batched GEMMs + fused batch-norm/ReLU/dropout passes (fused_bn_mlp.py)
            outs = fused_bn_mlp.run_stacks(box_features.reshape(-1, channel), parsed)
            raw = {n: out.view(num_layers, num_queries, batch, -1).permute(0, 2, 1, 3) for n, out in zip(names, outs)}
        else:
            feats = box_features.permute(0, 2, 3, 1).reshape(num_layers * batch, channel, num_queries)
            raw = {n: heads[n](feats).transpose(1, 2).reshape(num_layers, batch, num_queries, -1) for n in names}

        cls_logits = raw["sem_cls_head"]
        text_correlation_embedding = raw["text_correlation_head"]
        angle_logits = raw["angle_cls_head"]
        angle_residual_normalized = raw["angle_residual_head"]
        decode_in = (raw["center_head"], raw["size_head"], angle_logits, angle_residual_normalized, cls_logits)
        if box_decode.eligible(decode_in, query_xyz, point_cloud_dims, self.box_processor.dataset_config):
            # one kernel per direction for the whole element-wise tail below (csrc/box_decode.hip)
            dec = box_decode.decode(*decode_in, query_xyz, point_cloud_dims)
            stacked = dict(dec, sem_cls_logits=cls_logits, text_correlation_embedding=text_correlation_embedding,
                           angle_logits=angle_logits, angle_residual_normalized=angle_residual_normalized)
            return self._layer_dicts(stacked, num_layers, point_clouds)
        center_offset = raw["center_head"].sigmoid() - 0.5
        size_normalized = raw["size_head"].sigmoid()
        angle_residual = angle_residual_normalized * (np.pi / angle_residual_normalized.shape[-1])

        # Box decoding (models/model_3detr.py:1683-1731).  The reference decodes the num_layers
        # decoder outputs one after the other (~70 tiny kernels per layer, forward and again in
        # backward); the same elementwise arithmetic is applied here to all layers at once
        # (leading dim num_layers*batch, layer-major) and sliced per layer afterwards.
        nlb = num_layers * batch

        def flat(t):
            return t.reshape(nlb, num_queries, -1)

        dims_rep = [d.repeat(num_layers, 1) for d in point_cloud_dims]
        query_rep = query_xyz.repeat(num_layers, 1, 1)
        center_normalized, center_unnormalized = self.box_processor.compute_predicted_center(
            flat(center_offset), query_rep, dims_rep)
        angle_continuous = self.box_processor.compute_predicted_angle(flat(angle_logits),
                                                                      flat(angle_residual))
        size_unnormalized = self.box_processor.compute_predicted_size(flat(size_normalized), dims_rep)
        box_corners = self.box_processor.box_parametrization_to_corners(
            center_unnormalized, size_unnormalized, angle_continuous)
        box_corners_xyz = self.box_processor.box_parametrization_to_corners_xyz(
            center_unnormalized, size_unnormalized, angle_continuous)
        with torch.no_grad():  # matching / mAP only
            semcls_prob, objectness_prob = self.box_processor.compute_objectness_and_cls_prob(
                flat(cls_logits))

        def per_layer(t):
            return t.reshape(num_layers, batch, *t.shape[1:])

        center_normalized = per_layer(center_normalized.contiguous())
        center_unnormalized = per_layer(center_unnormalized)
        angle_continuous = per_layer(angle_continuous)
        size_unnormalized = per_layer(size_unnormalized)
        box_corners = per_layer(box_corners)
        box_corners_xyz = per_layer(box_corners_xyz)
        semcls_prob = per_layer(semcls_prob)
        objectness_prob = per_layer(objectness_prob)

        stacked = {
            "sem_cls_logits": cls_logits,
            "text_correlation_embedding": text_correlation_embedding,
            "center_normalized": center_normalized,
            "center_unnormalized": center_unnormalized,
            "size_normalized": size_normalized,
            "size_unnormalized": size_unnormalized,
            "angle_logits": angle_logits,
            "angle_residual": angle_residual,
            "angle_residual_normalized": angle_residual_normalized,
            "angle_continuous": angle_continuous,
            "objectness_prob": objectness_prob,
            "sem_cls_prob": semcls_prob,
            "box_corners": box_corners,
            "box_corners_xyz": box_corners_xyz,
        }
        return self._layer_dicts(stacked, num_layers, point_clouds)

    @staticmethod
    def _layer_dicts(stacked, num_layers, point_clouds):
        outputs = [dict({k: v[l] for k, v in stacked.items()}, point_clouds=point_clouds)
                   for l in range(num_layers)]
        # `stacked_outputs` ((num_layers, batch, ...) per key, last decoder layer at index -1) is this
        # package's addition to the reference's return value: criterion.SetCriterion evaluates all
        # layers from it in one pass instead of looping over outputs / aux_outputs.
        return {"outputs": outputs[-1], "aux_outputs": outputs[:-1], "stacked_outputs": stacked}

    def get_class_scores(self, box_predictions):
        """Open-vocabulary class scores of the evaluated decoder layer: softmax over the prompts of
        (unit-norm region embedding . text embedding) * temperature.  Overwrites ``sem_cls_prob``."""
        if self.eval_layer_id != -1:  # evaluate an intermediate decoder layer instead of the last one
            box_predictions["outputs"].update(box_predictions["aux_outputs"][self.eval_layer_id])
        out = box_predictions["outputs"]
        region = out["text_correlation_embedding"]
        region = region / (region.norm(dim=-1, keepdim=True) + 1e-32)
        prompts = out["text_features_clip"].to(torch.float32)
        logits = torch.bmm(region, prompts.transpose(1, 2)) * out["logit_scale"]
        out["sem_cls_prob"] = torch.softmax(logits, dim=-1)
        return box_predictions, out["sem_cls_prob"], out["objectness_prob"]

    @_lib.on_tensor_device(lambda inputs, *a, **k: inputs.get("point_clouds"))
    def forward(self, inputs, encoder_only=False, if_test=False, if_real_test=False, curr_epoch=-1,
                if_cmp_class=False, pre_encoded=None):
        """models/model_3detr.py:1767-1817.  ``pre_encoded`` (this package's addition): the result of
        ``run_pre_encoder`` on ``inputs["point_clouds"]``, when the caller ran that stage itself."""
        point_clouds = inputs["point_clouds"]
        enc_xyz, enc_features, enc_inds = self.run_encoder(point_clouds, pre_encoded)
        enc_features = self._project_encoder_features(enc_features)
        if encoder_only:
            return enc_xyz, enc_features.transpose(0, 1)
        point_cloud_dims = [inputs["point_cloud_dims_min"], inputs["point_cloud_dims_max"]]
        query_xyz, query_embed = self.get_query_embeddings(enc_xyz, point_cloud_dims)
        enc_pos = self.pos_embedding(enc_xyz, input_range=point_cloud_dims)
        enc_pos = enc_pos.permute(2, 0, 1)
        query_embed = query_embed.permute(2, 0, 1)
        tgt = torch.zeros_like(query_embed)
        box_features = self.decoder(tgt, enc_features, query_pos=query_embed, pos=enc_pos)[0]
        box_predictions = self.get_box_predictions(query_xyz, point_cloud_dims, box_features,
                                                   point_clouds, inputs)
        if self.text_features_fg_norm is None:
            return box_predictions

        outputs = box_predictions["outputs"]
        outputs["logit_scale"] = torch.clip(self.logit_scale.exp(), min=None, max=100)
        bsz = point_clouds.shape[0]
        if (not if_real_test) and (not if_cmp_class) and (not if_test):  # :1799-1817
            prompts = self.superset_text_features_fg_norm if self.if_clip_superset \
                else self.text_features_fg_norm[:self.train_range_max, :]
            # (B, ncls, E) as a VIEW (the reference materialises B copies, models/model_3detr.py:1806: same values): the
            # criterion can then see that all scenes share one prompt set and run the class logits as one dense product
            outputs["text_features_clip"] = prompts.unsqueeze(0).expand(bsz, -1, -1)
            provider = self.region_embedding_provider
            if provider is not None:
                if getattr(provider, "stage2", self.online_nms_update_save_novel_label_clip_driven_with_cate_confidence):
                    outputs["maybe_novel_text_features_clip"] = self.superset_text_features_fg_norm \
                        if self.if_clip_superset else self.text_features_fg_norm[:self.test_range_max, :]
                if hasattr(provider, "if_keep_box"):
                    provider.if_keep_box = self.if_keep_box   # main.py:356 sets it on the model at epoch boundaries
                box_predictions["outputs"] = provider(inputs, outputs, curr_epoch=curr_epoch)
        if if_real_test:
            outputs["text_features_clip"] = self.text_features_fg_norm.unsqueeze(0).repeat(bsz, 1, 1)
            box_predictions, _, _ = self.get_class_scores(box_predictions)
        return box_predictions


# ---- builders (models/model_3detr.py:3935-3996, 4018-4048; models/__init__.py:3-10) ----------
def build_preencoder(args):
    mlp_dims = [3 * int(args.use_color), 64, 128, args.enc_dim]
    return PointnetSAModuleVotes(radius=0.2, nsample=64, npoint=args.preenc_npoints, mlp=mlp_dims,
                                 normalize_xyz=True)


def build_encoder(args):
    if args.enc_type == "vanilla":
        encoder_layer = TransformerEncoderLayer(d_model=args.enc_dim, nhead=args.enc_nhead,
                                                dim_feedforward=args.enc_ffn_dim,
                                                dropout=args.enc_dropout,
                                                activation=args.enc_activation)
        return TransformerEncoder(encoder_layer=encoder_layer, num_layers=args.enc_nlayers)
    if args.enc_type in ["masked"]:
        encoder_layer = TransformerEncoderLayer(d_model=args.enc_dim, nhead=args.enc_nhead,
                                                dim_feedforward=args.enc_ffn_dim,
                                                dropout=args.enc_dropout,
                                                activation=args.enc_activation)
        interim_downsampling = PointnetSAModuleVotes(radius=0.4, nsample=32,
                                                     npoint=args.preenc_npoints // 2,
                                                     mlp=[args.enc_dim, 256, 256, args.enc_dim],
                                                     normalize_xyz=True)
        masking_radius = [math.pow(x, 2) for x in [0.4, 0.8, 1.2]]
        return MaskedTransformerEncoder(encoder_layer=encoder_layer, num_layers=3,
                                        interim_downsampling=interim_downsampling,
                                        masking_radius=masking_radius)
    raise ValueError(f"Unknown encoder type {args.enc_type}")


def build_decoder(args):
    decoder_layer = TransformerDecoderLayer(d_model=args.dec_dim, nhead=args.dec_nhead,
                                            dim_feedforward=args.dec_ffn_dim,
                                            dropout=args.dec_dropout)
    return TransformerDecoder(decoder_layer, num_layers=args.dec_nlayers, return_intermediate=True)


# Optional hook: callable(args, dataset_config) -> dict with any of `text_features_fg_norm`,
# `logit_scale`, `clip_model`, `region_embedding_provider`.  A deployment next to the reference sets it
# once so that main.py's `build_model(args, dataset_config)` call is unchanged.
CLIP_LOADER = None


def build_3detr_predictedbox_distillation_head(args, dataset_config, **extra):
    if CLIP_LOADER is not None and "text_features_fg_norm" not in extra:
        extra = dict(CLIP_LOADER(args, dataset_config), **extra)
    pre_encoder = build_preencoder(args)
    encoder = build_encoder(args)
    decoder = build_decoder(args)
    g = lambda k, d=False: getattr(args, k, d)  # noqa: E731  (flags of main.py:37-304)
    model = Model3DETRPredictedBoxDistillationHead(
        pre_encoder, encoder, decoder, dataset_config, encoder_dim=args.enc_dim,
        decoder_dim=args.dec_dim, mlp_dropout=args.mlp_dropout, num_queries=args.nqueries,
        if_with_clip=g("if_with_clip"), if_with_clip_embed=g("if_with_clip_embed"),
        if_use_gt_box=g("if_use_gt_box"), if_expand_box=g("if_expand_box"),
        if_with_fake_classes=g("if_with_fake_classes"), pooling_methods=g("pooling_methods", "average"),
        if_clip_more_prompts=g("if_clip_more_prompts"), if_keep_box=g("if_keep_box"),
        if_select_box_by_objectness=g("if_select_box_by_objectness"),
        keep_objectness=g("keep_objectness", 0.5),
        online_nms_update_novel_label=g("online_nms_update_novel_label"),
        online_nms_update_accumulate_novel_label=g("online_nms_update_accumulate_novel_label"),
        online_nms_update_accumulate_epoch=g("online_nms_update_accumulate_epoch", 10),
        distillation_box_num=g("distillation_box_num", 32), args=args, **extra)
    return model, BoxProcessor(dataset_config)


MODEL_FUNCS = {
    "3detr_predictedbox_distillation": build_3detr_predictedbox_distillation_head,
}


def build_model(args, dataset_config, **extra):
    """models/__init__.py:8-10."""
    return MODEL_FUNCS[args.model_name](args, dataset_config, **extra)


def default_args(**overrides):
    """The hot-path flags with main.py's defaults (main.py:64-74,127-148)."""
    from types import SimpleNamespace
    ns = SimpleNamespace(model_name="3detr_predictedbox_distillation", use_color=False, enc_type="vanilla",
                         enc_nlayers=3, enc_dim=256, enc_ffn_dim=128, enc_dropout=0.1, enc_nhead=4,
                         enc_activation="relu", dec_nlayers=8, dec_dim=256, dec_ffn_dim=256,
                         dec_dropout=0.1, dec_nhead=4, mlp_dropout=0.3, preenc_npoints=2048,
                         nqueries=256, eval_layer_id=-1, dataset_name="sunrgbd")
    for k, v in overrides.items():
        setattr(ns, k, v)
    return ns
