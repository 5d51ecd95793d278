"""Small building blocks shared by the transformer and the model heads.

Mirror of models/helpers.py:1-116: ``GenericMLP`` (Linear/Conv1d + norm +
activation + dropout stacks whose ``state_dict`` keys are ``layers.{i}.*``),
``BatchNormDim1Swap``, the NORM/ACTIVATION/WEIGHT_INIT registries and
``get_clones``.
"""
import copy
from functools import partial

import torch.nn as nn


class BatchNormDim1Swap(nn.BatchNorm1d):
    """BatchNorm1d over the channel dim of a (HW, N, C) sequence-first tensor
    (models/helpers.py:7-24)."""

    def forward(self, x):
        x = x.permute(1, 2, 0)  # (N, C, HW)
        x = super().forward(x)
        return x.permute(2, 0, 1)


NORM_DICT = {
    "bn": BatchNormDim1Swap,
    "bn1d": nn.BatchNorm1d,
    "id": nn.Identity,
    "ln": nn.LayerNorm,
}

ACTIVATION_DICT = {
    "relu": nn.ReLU,
    "gelu": nn.GELU,
    "leakyrelu": partial(nn.LeakyReLU, negative_slope=0.1),
}

WEIGHT_INIT_DICT = {
    "xavier_uniform": nn.init.xavier_uniform_,
}


class GenericMLP(nn.Module):
    """models/helpers.py:45-112.  ``use_conv`` selects Conv1d(k=1) over Linear;
    with ``norm_fn_name='ln'`` and conv layers the norm is GroupNorm(1, C)."""

    def __init__(self, input_dim, hidden_dims, output_dim, norm_fn_name=None, activation="relu",
                 use_conv=False, dropout=None, hidden_use_bias=False, output_use_bias=True,
                 output_use_activation=False, output_use_norm=False, weight_init_name=None):
        super().__init__()
        act = ACTIVATION_DICT[activation]
        norm = NORM_DICT[norm_fn_name] if norm_fn_name is not None else None
        if norm_fn_name == "ln" and use_conv:
            norm = lambda c: nn.GroupNorm(1, c)  # noqa: E731
        if dropout is not None and not isinstance(dropout, list):
            dropout = [dropout for _ in range(len(hidden_dims))]

        def dense(cin, cout, bias):
            return nn.Conv1d(cin, cout, 1, bias=bias) if use_conv else nn.Linear(cin, cout, bias=bias)

        layers = []
        prev = input_dim
        for i, width in enumerate(hidden_dims):
            layers.append(dense(prev, width, hidden_use_bias))
            if norm:
                layers.append(norm(width))
            layers.append(act())
            if dropout is not None:
                layers.append(nn.Dropout(p=dropout[i]))
            prev = width
        layers.append(dense(prev, output_dim, output_use_bias))
        if output_use_norm:
            layers.append(norm(output_dim))
        if output_use_activation:
            layers.append(act())
        self.layers = nn.Sequential(*layers)
        if weight_init_name is not None:
            self.do_weight_init(weight_init_name)

    def do_weight_init(self, weight_init_name):
        func = WEIGHT_INIT_DICT[weight_init_name]
        for _, param in self.named_parameters():
            if param.dim() > 1:  # skips norm scales / biases
                func(param)

    def forward(self, x):
        return self.layers(x)

    def tokens_supported(self):
        """True when the stack is token-wise without batch statistics (k=1 convolutions or
        Linear, activations, dropout): it can then run on channels-last (..., C) tokens."""
        for m in self.layers:
            if isinstance(m, nn.Conv1d):
                if m.kernel_size != (1,) or m.stride != (1,) or m.padding != (0,) or m.groups != 1:
                    return False
            elif not isinstance(m, (nn.Linear, nn.ReLU, nn.GELU, nn.LeakyReLU, nn.Dropout, nn.Identity)):
                return False
        return True

    def forward_tokens(self, x):
        """Same function as ``forward`` on (..., C) tokens instead of (N, C, tokens): a k=1
        Conv1d is a Linear over the channel axis (one library GEMM, no layout transposes)."""
        for m in self.layers:
            if isinstance(m, nn.Conv1d):
                x = nn.functional.linear(x, m.weight.squeeze(-1), m.bias)
            else:
                x = m(x)
        return x


def get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])
