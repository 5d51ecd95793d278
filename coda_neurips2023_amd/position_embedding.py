"""Coordinate position embeddings of the 3DETR decoder.

API and checkpoint contract of models/position_embedding.py:12-139: a module
``PositionEmbeddingCoordsSine(d_pos=..., pos_type=..., normalize=...)`` called as
``emb(xyz (B,N,3), input_range=[min (B,3), max (B,3)]) -> (B, d_pos, N)`` under ``no_grad``,
whose random projection is the registered (checkpointed) buffer ``gauss_B (3, d_pos/2)``.
The model uses ``pos_type="fourier"``; the sinusoidal variant is kept for the API.
``shift_scale_points`` / ``scale_points`` are the two range helpers of utils/pc_util.py:38-73
that the embedding and the box decoder share.

Arithmetic follows the reference operation for operation (normalise, times 2*pi, project) so the
fp32 results are the same numbers; the code is written for this package.
"""
import math

import torch
from torch import nn


def _per_scene(t, like):
    """(B,3) per-scene vector -> broadcastable against `like` (B, ..., 3)."""
    return t.reshape(t.shape[0], *([1] * (like.ndim - 2)), t.shape[-1])


def shift_scale_points(pred_xyz, src_range, dst_range=None):
    """Map points from the box ``src_range = [lo, hi]`` onto ``dst_range`` (the unit cube when omitted):
    ((p - src_lo) * dst_extent) / src_extent + dst_lo, per scene."""
    src_lo, src_hi = (_per_scene(t, pred_xyz) for t in src_range)
    if src_lo.shape[0] != pred_xyz.shape[0] or src_lo.shape[-1] != pred_xyz.shape[-1]:
        raise ValueError("src_range does not match the points (batch size / dimensionality)")
    offset = pred_xyz - src_lo
    if dst_range is None:
        return offset / (src_hi - src_lo)  # extent 1 and origin 0 change no bits
    dst_lo, dst_hi = (_per_scene(t, pred_xyz) for t in dst_range)
    return offset * (dst_hi - dst_lo) / (src_hi - src_lo) + dst_lo


def scale_points(pred_xyz, mult_factor):
    """Per-scene, per-axis scaling (box sizes back to metres)."""
    return pred_xyz * _per_scene(mult_factor, pred_xyz)


class PositionEmbeddingCoordsSine(nn.Module):
    def __init__(self, temperature=10000, normalize=False, scale=None, pos_type="fourier", d_pos=None, d_in=3,
                 gauss_scale=1.0):
        super().__init__()
        if pos_type not in ("sine", "fourier"):
            raise ValueError(f"Unknown {pos_type}")
        if scale is not None and not normalize:
            raise ValueError("normalize should be True if scale is passed")
        self.temperature = temperature
        self.normalize = normalize
        self.scale = 2 * math.pi if scale is None else scale
        self.pos_type = pos_type
        if pos_type == "fourier":
            if d_pos is None or d_pos % 2:
                raise ValueError("fourier embeddings need an even d_pos")
            self.d_pos = d_pos
            self.register_buffer("gauss_B", torch.randn(d_in, d_pos // 2) * gauss_scale)

    def _unit(self, xyz, input_range):
        return shift_scale_points(xyz, src_range=input_range) if self.normalize else xyz

    def get_fourier_embeddings(self, xyz, num_channels=None, input_range=None):
        """[sin, cos]((2*pi * xyz_unit) @ gauss_B[:, :num_channels/2]) -> (B, num_channels, N)."""
        half = self.gauss_B.shape[1] if num_channels is None else num_channels // 2
        if half <= 0 or half > self.gauss_B.shape[1] or (num_channels is not None and num_channels % 2):
            raise ValueError("num_channels must be even and at most 2 * gauss_B.shape[1]")
        if (xyz.is_cuda and xyz.dtype == torch.float32 and xyz.shape[-1] == 3 and self.gauss_B.dtype == torch.float32
                and self.gauss_B.is_cuda and not torch.is_grad_enabled()):
            # one launch instead of eight (csrc/pos_embed.hip, include/coda_token_ops.h part 3)
            from . import _lib
            b, n = xyz.shape[0], xyz.shape[1]
            pts = xyz.contiguous()
            lo = hi = None
            if self.normalize:
                lo = input_range[0].to(torch.float32).reshape(b, 3).contiguous()
                hi = input_range[1].to(torch.float32).reshape(b, 3).contiguous()
            gauss = self.gauss_B if self.gauss_B.stride(1) == 1 else self.gauss_B.contiguous()
            out = torch.empty((b, n, 2 * half), dtype=torch.float32, device=xyz.device)
            with torch.cuda.device(xyz.device):
                st = _lib.load().coda_fourier_pos_embed_f32(pts.data_ptr(), lo.data_ptr() if lo is not None else None,
                                                            hi.data_ptr() if hi is not None else None, gauss.data_ptr(),
                                                            gauss.stride(0), out.data_ptr(), b, n, half,
                                                            _lib.current_stream_handle())
            _lib.check(st, "coda_fourier_pos_embed_f32")
            return out.transpose(1, 2)
        phase = (self._unit(xyz, input_range) * (2 * math.pi)).reshape(-1, xyz.shape[-1]) @ self.gauss_B[:, :half]
        phase = phase.view(xyz.shape[0], xyz.shape[1], half)
        return torch.cat((phase.sin(), phase.cos()), dim=2).transpose(1, 2)

    def get_sine_embeddings(self, xyz, num_channels, input_range):
        """Transformer-style sinusoids per coordinate.  The channels are split evenly over the
        coordinates in even counts; what is left over goes to the leading coordinates two at a time.
        Channel c of a coordinate's block is sin (c even) or cos (c odd) of x * scale / T^(2*(c//2)/width)."""
        ncoord = xyz.shape[2]
        base = (num_channels // ncoord) // 2 * 2
        spare = num_channels - base * ncoord
        unit = self._unit(xyz, input_range)
        blocks = []
        for axis in range(ncoord):
            width = base + (2 if spare > 2 * axis else 0)
            c = torch.arange(width, dtype=torch.float32, device=xyz.device)
            wavelength = self.temperature ** (2 * torch.div(c, 2, rounding_mode="floor") / width)
            arg = (unit[:, :, axis] * self.scale if self.scale else unit[:, :, axis]).unsqueeze(-1) / wavelength
            blocks.append(torch.where(c.long() % 2 == 0, arg.sin(), arg.cos()))
        return torch.cat(blocks, dim=2).transpose(1, 2)

    def forward(self, xyz, num_channels=None, input_range=None):
        if not torch.is_tensor(xyz) or xyz.ndim != 3:
            raise ValueError("xyz must be a (B, N, 3) tensor")
        with torch.no_grad():
            if self.pos_type == "fourier":
                return self.get_fourier_embeddings(xyz, num_channels, input_range)
            return self.get_sine_embeddings(xyz, num_channels, input_range)

    def extra_repr(self):
        rep = f"type={self.pos_type}, scale={self.scale}, normalize={self.normalize}"
        if hasattr(self, "gauss_B"):
            rep += f", gaussB={tuple(self.gauss_B.shape)}"
        return rep
