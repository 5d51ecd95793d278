"""Set-abstraction / feature-propagation modules.

The two classes of third_party_pointnet2/pointnet2/pointnet2_modules.py that the CoDA models build:
``PointnetSAModuleVotes`` (:161-268: the pre-encoder and the masked encoder's interim down-sampling,
models/model_3detr.py:3935-3972) and ``PointnetFPModule`` (:352-411, the three_nn / three_interpolate
consumer).  Class names, constructor keywords, forward signature / returns and ``state_dict`` keys
(``mlp_module.layer{i}...``, ``mlp.layer{i}...``) are the reference's; the bodies are this package's.
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from . import _ext, fused_sa_mlp, pointnet2_utils
from . import pytorch_utils as pt_utils


class PointnetSAModuleVotes(nn.Module):
    """One set-abstraction level: sample ``npoint`` centres (FPS), group ``nsample`` neighbours within
    ``radius`` of each, run the shared MLP over every group and pool it; the sampled indices are returned as
    well ("votes" variant).  ``npoint=None`` groups the whole cloud into one region."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pooling: str = "max", sigma: float = None,
                 normalize_xyz: bool = False, sample_uniformly: bool = False, ret_unique_cnt: bool = False):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.pooling, self.use_xyz = pooling, use_xyz
        self.normalize_xyz, self.ret_unique_cnt = normalize_xyz, ret_unique_cnt
        self.sigma = sigma if sigma is not None else (radius / 2 if radius is not None else None)  # "rbf" pooling width
        if npoint is None:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)
        else:
            self.grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                                                         normalize_xyz=normalize_xyz,
                                                         sample_uniformly=sample_uniformly,
                                                         ret_unique_cnt=ret_unique_cnt)
        if use_xyz and mlp:
            mlp[0] += 3  # the caller's list grows by the xyz channels IN PLACE, as with the reference's module
        self.mlp_module = pt_utils.SharedMLP(mlp, bn=bn)

    def _fused(self, xyz, features):
        return (self.npoint is not None and not self.ret_unique_cnt and not self.grouper.sample_uniformly
                and not xyz.requires_grad
                and fused_sa_mlp.eligible(self.mlp_module, features, self.use_xyz, self.pooling, xyz))

    def prepare(self, xyz: torch.Tensor):
        """The parameter-free front of ``forward`` for an xyz-only module -- sampling, gathering
        the centres, ball query + grouping, distinct-row counts -- on the CURRENT stream, with the
        row count on its way to pinned host memory.  ``pointnet2_utils.SamplingPrefetcher`` runs
        this for an upcoming batch on a side stream; ``forward(..., prepared=...)`` continues from
        it.  Returns None when the module's configuration is not the fused xyz-only one."""
        if not self._fused(xyz, None):
            return None
        inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        idx, grouped_cl = _ext.query_and_group_xyz(new_xyz, xyz, self.radius, self.nsample,
                                                   self.normalize_xyz, channels_last=True)
        front = {"xyz": xyz, "inds": inds, "new_xyz": new_xyz, "idx": idx, "grouped_cl": grouped_cl}
        if fused_sa_mlp.mfma_eligible(self.mlp_module, self.nsample):
            # MFMA path: groups packed on the device, nothing goes to the host
            import os
            front["packed"] = fused_sa_mlp.pack_groups(idx, grouped_cl, fused_sa_mlp.mfma_widths(self.mlp_module),
                                                       dedup=os.environ.get("CODA_SA_DEDUP", "auto") != "0")
            front["total_host"] = None
            return front
        cnt, goff, tot = fused_sa_mlp.count_distinct_rows(idx)
        total_host = torch.empty(1, dtype=torch.int64, pin_memory=True)
        total_host.copy_(tot, non_blocking=True)
        front.update(counts=(cnt, goff, tot), total_host=total_host)
        return front

    @_lib.on_tensor_device()
    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None,
                inds: torch.Tensor = None, prepared: dict = None):
        """xyz (B,N,3), features (B,C,N) or None, inds (B,npoint) or None ->
        new_xyz (B,npoint,3), new_features (B,mlp[-1],npoint), inds (B,npoint)
        [, unique_cnt].  ``prepared``: the result of ``prepare`` on this very xyz (its host-side
        row count must have arrived, i.e. the stream it ran on has been waited for)."""
        if prepared is not None and features is None and self._fused(xyz, None):
            b, npoint = prepared["new_xyz"].shape[0], prepared["new_xyz"].shape[1]
            if "packed" in prepared:
                pooled = fused_sa_mlp.mfma_mlp_pool(prepared["packed"], b * npoint, self.nsample, self.mlp_module)
                return prepared["new_xyz"], pooled.view(b, npoint, -1).permute(0, 2, 1), prepared["inds"]
            pooled = fused_sa_mlp.fused_mlp_pool(prepared["grouped_cl"].view(-1, 3), b * npoint, self.nsample,
                                                 self.mlp_module, idx=prepared["idx"], counts=prepared["counts"],
                                                 total=None if prepared["total_host"] is None
                                                 else int(prepared["total_host"][0]))
            return prepared["new_xyz"], pooled.view(b, npoint, -1).permute(0, 2, 1), prepared["inds"]
        if inds is None:
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        elif inds.shape[1] != self.npoint:
            raise AssertionError("inds must hold npoint indices per scene")
        new_xyz = None
        if self.npoint is not None:  # (B,N,3) -> (B,3,N) -> gather -> (B,npoint,3)
            new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()

        if self._fused(xyz, features):
            # xyz-only set abstraction (the model's pre-encoder): fused ball query + grouping into
            # channels-last, then the fused shared MLP + batch norm + ReLU + max-pool
            idx, grouped_cl = _ext.query_and_group_xyz(new_xyz, xyz, self.radius, self.nsample,
                                                       self.normalize_xyz, channels_last=True)
            b, npoint = new_xyz.shape[0], new_xyz.shape[1]
            pooled = fused_sa_mlp.fused_mlp_pool(grouped_cl.view(-1, 3), b * npoint, self.nsample,
                                                 self.mlp_module, idx=idx)
            return new_xyz, pooled.view(b, npoint, -1).permute(0, 2, 1), inds  # (B, mlp[-1], npoint)

        grouped = self.grouper(xyz, new_xyz, features)  # (features, xyz[, distinct counts])
        per_sample = self.mlp_module(grouped[0])        # (B, mlp[-1], npoint, nsample)
        new_features = self._pool(per_sample, grouped[1]).squeeze(-1)
        return (new_xyz, new_features, inds) + tuple(grouped[2:])

    def _pool(self, per_sample, grouped_xyz):
        """(B,C,npoint,nsample) -> (B,C,npoint,1) by the configured pooling."""
        window = [1, per_sample.shape[3]]
        if self.pooling == "max":
            return F.max_pool2d(per_sample, kernel_size=window)
        if self.pooling == "avg":
            return F.avg_pool2d(per_sample, kernel_size=window)
        if self.pooling == "rbf":  # Gaussian of the neighbour's distance to its centre, averaged over the samples
            kernel = torch.exp(-grouped_xyz.pow(2).sum(1) / (self.sigma ** 2) / 2)
            return (per_sample * kernel.unsqueeze(1)).sum(-1, keepdim=True) / float(self.nsample)
        return per_sample


class PointnetFPModule(nn.Module):
    """Feature propagation: every ``unknown`` point takes the inverse-distance weighted mean of the features
    of its three nearest ``known`` points (``three_nn`` / ``three_interpolate``), concatenated with its own
    features if any, through a shared MLP.  ``known=None`` broadcasts one global feature."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    @_lib.on_tensor_device()
    def forward(self, unknown: torch.Tensor, known: torch.Tensor, unknow_feats: torch.Tensor,
                known_feats: torch.Tensor) -> torch.Tensor:
        if known is None:
            carried = known_feats.expand(known_feats.shape[0], known_feats.shape[1], unknown.shape[1])
        else:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            inverse = 1.0 / (dist + 1e-8)
            carried = pointnet2_utils.three_interpolate(known_feats, idx, inverse / inverse.sum(dim=2, keepdim=True))
        stacked = carried if unknow_feats is None else torch.cat((carried, unknow_feats), dim=1)
        return self.mlp(stacked.unsqueeze(-1)).squeeze(-1)
