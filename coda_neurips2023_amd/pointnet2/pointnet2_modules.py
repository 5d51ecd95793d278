"""Set-abstraction / feature-propagation modules.

Mirror of third_party_pointnet2/pointnet2/pointnet2_modules.py for the classes
the CoDA models build: ``PointnetSAModuleVotes`` (:161-268, the pre-encoder and
the masked encoder's interim down-sampling, models/model_3detr.py:3935-3972) and
``PointnetFPModule`` (:352-411, the three_nn / three_interpolate consumer).
Class names, constructor keywords, forward signature/returns and ``state_dict``
keys are those of the reference.
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _ext, fused_sa_mlp, pointnet2_utils
from . import pytorch_utils as pt_utils


class PointnetSAModuleVotes(nn.Module):
    """FPS -> gather -> ball-query grouping -> shared MLP -> pooling, returning
    the sampled indices as well (pointnet2_modules.py:161-268)."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None,
                 nsample: int = None, bn: bool = True, use_xyz: bool = True,
                 pooling: str = "max", sigma: float = None, normalize_xyz: bool = False,
                 sample_uniformly: bool = False, ret_unique_cnt: bool = False):
        super().__init__()
        self.npoint = npoint
        self.radius = radius
        self.nsample = nsample
        self.pooling = pooling
        self.mlp_module = None
        self.use_xyz = use_xyz
        self.sigma = sigma
        if self.sigma is None:
            self.sigma = self.radius / 2
        self.normalize_xyz = normalize_xyz
        self.ret_unique_cnt = ret_unique_cnt

        if npoint is not None:
            self.grouper = pointnet2_utils.QueryAndGroup(
                radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                normalize_xyz=normalize_xyz, sample_uniformly=sample_uniformly,
                ret_unique_cnt=ret_unique_cnt)
        else:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)

        mlp_spec = mlp
        if use_xyz and len(mlp_spec) > 0:
            mlp_spec[0] += 3  # in place, like the reference (:201-203)
        self.mlp_module = pt_utils.SharedMLP(mlp_spec, bn=bn)

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
        cnt, goff, tot = fused_sa_mlp.count_distinct_rows(idx)
        total_host = torch.empty(1, dtype=torch.int64, pin_memory=True)
        total_host.copy_(tot, non_blocking=True)
        return {"xyz": xyz, "inds": inds, "new_xyz": new_xyz, "idx": idx, "grouped_cl": grouped_cl,
                "counts": (cnt, goff, tot), "total_host": total_host}

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None,
                inds: torch.Tensor = None, prepared: dict = None):
        """xyz (B,N,3), features (B,C,N) or None, inds (B,npoint) or None ->
        new_xyz (B,npoint,3), new_features (B,mlp[-1],npoint), inds (B,npoint)
        [, unique_cnt].  ``prepared``: the result of ``prepare`` on this very xyz (its host-side
        row count must have arrived, i.e. the stream it ran on has been waited for)."""
        if prepared is not None and features is None and self._fused(xyz, None):
            b, npoint = prepared["new_xyz"].shape[0], prepared["new_xyz"].shape[1]
            pooled = fused_sa_mlp.fused_mlp_pool(prepared["grouped_cl"].view(-1, 3), b * npoint, self.nsample,
                                                 self.mlp_module, idx=prepared["idx"], counts=prepared["counts"],
                                                 total=None if prepared["total_host"] is None
                                                 else int(prepared["total_host"][0]))
            return prepared["new_xyz"], pooled.view(b, npoint, -1).permute(0, 2, 1), prepared["inds"]
        if inds is None:
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        else:
            assert inds.shape[1] == self.npoint
        if self.npoint is not None:
            xyz_flipped = xyz.transpose(1, 2).contiguous()
            new_xyz = pointnet2_utils.gather_operation(xyz_flipped, inds)
            new_xyz = new_xyz.transpose(1, 2).contiguous()
        else:
            new_xyz = None

        if self._fused(xyz, features):
            # xyz-only set abstraction (the model's pre-encoder): fused ball query + grouping into
            # channels-last, then the fused shared MLP + batch norm + ReLU + max-pool
            idx, grouped_cl = _ext.query_and_group_xyz(new_xyz, xyz, self.radius, self.nsample,
                                                       self.normalize_xyz, channels_last=True)
            b, npoint = new_xyz.shape[0], new_xyz.shape[1]
            pooled = fused_sa_mlp.fused_mlp_pool(grouped_cl.view(-1, 3), b * npoint, self.nsample,
                                                 self.mlp_module, idx=idx)
            new_features = pooled.view(b, npoint, -1).permute(0, 2, 1)  # (B, mlp[-1], npoint)
            return new_xyz, new_features, inds

        if not self.ret_unique_cnt:
            grouped_features, grouped_xyz = self.grouper(xyz, new_xyz, features)
        else:
            grouped_features, grouped_xyz, unique_cnt = self.grouper(xyz, new_xyz, features)

        new_features = self.mlp_module(grouped_features)  # (B, mlp[-1], npoint, nsample)
        if self.pooling == "max":
            new_features = F.max_pool2d(new_features, kernel_size=[1, new_features.size(3)])
        elif self.pooling == "avg":
            new_features = F.avg_pool2d(new_features, kernel_size=[1, new_features.size(3)])
        elif self.pooling == "rbf":
            # radial-basis weighted sum over the samples (:254-258)
            rbf = torch.exp(-1 * grouped_xyz.pow(2).sum(1, keepdim=False) / (self.sigma ** 2) / 2)
            new_features = torch.sum(new_features * rbf.unsqueeze(1), -1, keepdim=True) / float(
                self.nsample)
        new_features = new_features.squeeze(-1)  # (B, mlp[-1], npoint)

        if not self.ret_unique_cnt:
            return new_xyz, new_features, inds
        return new_xyz, new_features, inds, unique_cnt


class PointnetFPModule(nn.Module):
    """Feature propagation: inverse-distance interpolation from the 3 nearest
    known points, then a shared MLP (pointnet2_modules.py:352-411)."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    def forward(self, unknown: torch.Tensor, known: torch.Tensor, unknow_feats: torch.Tensor,
                known_feats: torch.Tensor) -> torch.Tensor:
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            norm = torch.sum(dist_recip, dim=2, keepdim=True)
            weight = dist_recip / norm
            interpolated_feats = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated_feats = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))

        if unknow_feats is not None:
            new_features = torch.cat([interpolated_feats, unknow_feats], dim=1)
        else:
            new_features = interpolated_feats
        new_features = self.mlp(new_features.unsqueeze(-1))
        return new_features.squeeze(-1)
