"""Coordinate position embeddings.

Mirror of models/position_embedding.py:12-139 (``PositionEmbeddingCoordsSine``)
and of the two point helpers it and the box decoder use
(utils/pc_util.py:38-73 ``shift_scale_points`` / ``scale_points``).  The Fourier
buffer ``gauss_B`` is registered (and checkpointed) under the same name.
"""
import math

import numpy as np
import torch
from torch import nn


def shift_scale_points(pred_xyz, src_range, dst_range=None):
    """Affine map of (B,N,3) points from ``src_range`` [min (B,3), max (B,3)] to
    ``dst_range`` (default the unit cube).  utils/pc_util.py:38-66."""
    if dst_range is None:
        dst_range = [
            torch.zeros((src_range[0].shape[0], 3), device=src_range[0].device),
            torch.ones((src_range[0].shape[0], 3), device=src_range[0].device),
        ]
    if pred_xyz.ndim == 4:
        src_range = [x[:, None] for x in src_range]
        dst_range = [x[:, None] for x in dst_range]
    assert src_range[0].shape[0] == pred_xyz.shape[0]
    assert dst_range[0].shape[0] == pred_xyz.shape[0]
    assert src_range[0].shape[-1] == pred_xyz.shape[-1]
    assert src_range[0].shape == src_range[1].shape
    assert dst_range[0].shape == dst_range[1].shape
    assert src_range[0].shape == dst_range[1].shape
    src_diff = src_range[1][:, None, :] - src_range[0][:, None, :]
    dst_diff = dst_range[1][:, None, :] - dst_range[0][:, None, :]
    return (((pred_xyz - src_range[0][:, None, :]) * dst_diff) / src_diff) + dst_range[0][:, None, :]


def scale_points(pred_xyz, mult_factor):
    """utils/pc_util.py:69-73."""
    if pred_xyz.ndim == 4:
        mult_factor = mult_factor[:, None]
    return pred_xyz * mult_factor[:, None, :]


class PositionEmbeddingCoordsSine(nn.Module):
    def __init__(self, temperature=10000, normalize=False, scale=None, pos_type="fourier",
                 d_pos=None, d_in=3, gauss_scale=1.0):
        super().__init__()
        self.temperature = temperature
        self.normalize = normalize
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        if scale is None:
            scale = 2 * math.pi
        assert pos_type in ["sine", "fourier"]
        self.pos_type = pos_type
        self.scale = scale
        if pos_type == "fourier":
            assert d_pos is not None
            assert d_pos % 2 == 0
            B = torch.empty((d_in, d_pos // 2)).normal_()
            B *= gauss_scale
            self.register_buffer("gauss_B", B)
            self.d_pos = d_pos

    def get_sine_embeddings(self, xyz, num_channels, input_range):
        """models/position_embedding.py:41-87."""
        xyz = xyz.clone()
        if self.normalize:
            xyz = shift_scale_points(xyz, src_range=input_range)
        ndim = num_channels // xyz.shape[2]
        if ndim % 2 != 0:
            ndim -= 1
        rems = num_channels - (ndim * xyz.shape[2])  # remainder goes to the first dims, 2 at a time
        assert ndim % 2 == 0
        final_embeds = []
        prev_dim = 0
        for d in range(xyz.shape[2]):
            cdim = ndim
            if rems > 0:
                cdim += 2
                rems -= 2
            if cdim != prev_dim:
                dim_t = torch.arange(cdim, dtype=torch.float32, device=xyz.device)
                dim_t = self.temperature ** (2 * (dim_t // 2) / cdim)
            raw_pos = xyz[:, :, d]
            if self.scale:
                raw_pos *= self.scale
            pos = raw_pos[:, :, None] / dim_t
            pos = torch.stack((pos[:, :, 0::2].sin(), pos[:, :, 1::2].cos()), dim=3).flatten(2)
            final_embeds.append(pos)
            prev_dim = cdim
        return torch.cat(final_embeds, dim=2).permute(0, 2, 1)

    def get_fourier_embeddings(self, xyz, num_channels=None, input_range=None):
        """Random Fourier features: [sin, cos](2*pi * xyz_normalised @ gauss_B)
        -> (B, d_pos, N).  models/position_embedding.py:89-118."""
        if num_channels is None:
            num_channels = self.gauss_B.shape[1] * 2
        bsize, npoints = xyz.shape[0], xyz.shape[1]
        assert num_channels > 0 and num_channels % 2 == 0
        d_in, max_d_out = self.gauss_B.shape[0], self.gauss_B.shape[1]
        d_out = num_channels // 2
        assert d_out <= max_d_out
        assert d_in == xyz.shape[-1]
        xyz = xyz.clone()
        if self.normalize:
            xyz = shift_scale_points(xyz, src_range=input_range)
        xyz *= 2 * np.pi
        xyz_proj = torch.mm(xyz.view(-1, d_in), self.gauss_B[:, :d_out]).view(bsize, npoints, d_out)
        final_embeds = [xyz_proj.sin(), xyz_proj.cos()]
        return torch.cat(final_embeds, dim=2).permute(0, 2, 1)

    def forward(self, xyz, num_channels=None, input_range=None):
        assert isinstance(xyz, torch.Tensor)
        assert xyz.ndim == 3
        with torch.no_grad():
            if self.pos_type == "sine":
                return self.get_sine_embeddings(xyz, num_channels, input_range)
            if self.pos_type == "fourier":
                return self.get_fourier_embeddings(xyz, num_channels, input_range)
        raise ValueError(f"Unknown {self.pos_type}")

    def extra_repr(self):
        st = f"type={self.pos_type}, scale={self.scale}, normalize={self.normalize}"
        if hasattr(self, "gauss_B"):
            st += f", gaussB={self.gauss_B.shape}, gaussBsum={self.gauss_B.sum().item()}"
        return st
