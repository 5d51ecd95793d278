"""The shared point-wise MLP of a set-abstraction module.

What the reference's checkpoints pin down (third_party_pointnet2/pointnet2/pytorch_utils.py:8-33
as instantiated by pointnet2_modules.py:205-206) is a naming contract, not a class hierarchy:

    mlp_module.layer{i}.conv.weight                        (C_out, C_in, 1, 1), no bias when normed
    mlp_module.layer{i}.bn.bn.{weight, bias, running_mean, running_var, num_batches_tracked}

with each layer computing ReLU(BatchNorm2d(Conv2d_1x1(x))) on a (B, C, npoint, nsample) tensor.
This file provides exactly that.  The fused channels-last path (fused_sa_mlp.py) reads the same
parameters through ``layer.conv`` / ``layer.bn.bn`` and recognises a layer by its three children
``conv, bn, activation``.
"""
import torch.nn as nn


class _Norm(nn.Module):
    """BatchNorm2d held under the attribute ``bn`` (gives the ``bn.bn.*`` checkpoint keys).
    SyncBatchNorm conversion replaces the inner module in place."""

    def __init__(self, channels):
        super().__init__()
        self.bn = nn.BatchNorm2d(channels)  # affine, weight 1 / bias 0: torch's defaults

    def forward(self, x):
        return self.bn(x)


def _pointwise_layer(c_in, c_out, normed):
    layer = nn.Sequential()
    conv = nn.Conv2d(c_in, c_out, kernel_size=1, bias=not normed)
    nn.init.kaiming_normal_(conv.weight)  # the reference's initialisation (pytorch_utils.py:98,165)
    if conv.bias is not None:
        nn.init.zeros_(conv.bias)
    layer.add_module("conv", conv)
    if normed:
        layer.add_module("bn", _Norm(c_out))
    layer.add_module("activation", nn.ReLU(inplace=True))
    return layer


class SharedMLP(nn.Sequential):
    """``SharedMLP([c0, c1, ..., cn], bn=True)``: n point-wise layers named ``layer0 .. layer{n-1}``."""

    def __init__(self, widths, *, bn=False):
        super().__init__()
        for i, (c_in, c_out) in enumerate(zip(widths[:-1], widths[1:])):
            self.add_module(f"layer{i}", _pointwise_layer(c_in, c_out, bn))
