"""SharedMLP and its conv/BN building blocks.

Mirror of third_party_pointnet2/pointnet2/pytorch_utils.py:8-117 restricted to
what the set-abstraction path instantiates.  Module/attribute names are kept so
that ``state_dict`` keys match the reference checkpoints
(``mlp_module.layer{i}.conv.weight``, ``mlp_module.layer{i}.bn.bn.{weight,bias,
running_mean,running_var,num_batches_tracked}``).
"""
from typing import List

import torch.nn as nn


class _BNBase(nn.Sequential):
    """BatchNorm wrapped in a Sequential under the child name ``bn``
    (pytorch_utils.py:36-44): weight = 1, bias = 0 at init."""

    def __init__(self, in_size, batch_norm=None, name=""):
        super().__init__()
        self.add_module(name + "bn", batch_norm(in_size))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class BatchNorm1d(_BNBase):
    def __init__(self, in_size: int, *, name: str = ""):
        super().__init__(in_size, batch_norm=nn.BatchNorm1d, name=name)


class BatchNorm2d(_BNBase):
    def __init__(self, in_size: int, name: str = ""):
        super().__init__(in_size, batch_norm=nn.BatchNorm2d, name=name)


class _ConvBase(nn.Sequential):
    """conv (+ bn) (+ activation), or the pre-activation order when ``preact``
    (pytorch_utils.py:64-117).  The conv has a bias only when there is no BN."""

    def __init__(self, in_size, out_size, kernel_size, stride, padding, activation, bn, init,
                 conv=None, batch_norm=None, bias=True, preact=False, name=""):
        super().__init__()
        bias = bias and (not bn)
        conv_unit = conv(in_size, out_size, kernel_size=kernel_size, stride=stride,
                         padding=padding, bias=bias)
        init(conv_unit.weight)
        if bias:
            nn.init.constant_(conv_unit.bias, 0)
        bn_unit = None
        if bn:
            bn_unit = batch_norm(in_size if preact else out_size)
        if preact:
            if bn_unit is not None:
                self.add_module(name + "bn", bn_unit)
            if activation is not None:
                self.add_module(name + "activation", activation)
        self.add_module(name + "conv", conv_unit)
        if not preact:
            if bn_unit is not None:
                self.add_module(name + "bn", bn_unit)
            if activation is not None:
                self.add_module(name + "activation", activation)


class Conv1d(_ConvBase):
    def __init__(self, in_size: int, out_size: int, *, kernel_size: int = 1, stride: int = 1,
                 padding: int = 0, activation=nn.ReLU(inplace=True), bn: bool = False,
                 init=nn.init.kaiming_normal_, bias: bool = True, preact: bool = False,
                 name: str = ""):
        super().__init__(in_size, out_size, kernel_size, stride, padding, activation, bn, init,
                         conv=nn.Conv1d, batch_norm=BatchNorm1d, bias=bias, preact=preact,
                         name=name)


class Conv2d(_ConvBase):
    def __init__(self, in_size: int, out_size: int, *, kernel_size=(1, 1), stride=(1, 1),
                 padding=(0, 0), activation=nn.ReLU(inplace=True), bn: bool = False,
                 init=nn.init.kaiming_normal_, bias: bool = True, preact: bool = False,
                 name: str = ""):
        super().__init__(in_size, out_size, kernel_size, stride, padding, activation, bn, init,
                         conv=nn.Conv2d, batch_norm=BatchNorm2d, bias=bias, preact=preact,
                         name=name)


class SharedMLP(nn.Sequential):
    """Stack of 1x1 Conv2d(+BN)+ReLU named ``layer{i}`` (pytorch_utils.py:8-33)."""

    def __init__(self, args: List[int], *, bn: bool = False, activation=nn.ReLU(inplace=True),
                 preact: bool = False, first: bool = False, name: str = ""):
        super().__init__()
        for i in range(len(args) - 1):
            plain_first = first and preact and i == 0
            self.add_module(
                name + "layer{}".format(i),
                Conv2d(args[i], args[i + 1], bn=(not plain_first) and bn,
                       activation=None if plain_first else activation, preact=preact),
            )
