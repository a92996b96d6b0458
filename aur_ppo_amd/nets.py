"""Tanh MLP policy / value networks with the reference's constructor signatures and state-dict
layout (src/nets/nets.py:14-53): ``<net>.net.{0,2,4,...}.{weight,bias}``, orthogonal init with
gain sqrt(2) on hidden layers, 0.01 on the actor head, 1.0 on the critic head, zero biases.
These run as stock PyTorch-ROCm GEMMs (hipBLASLt); they are host code, not custom kernels."""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn


def layer_init(layer, std=math.sqrt(2), bias_const=0.0):
    nn.init.orthogonal_(layer.weight, std)
    nn.init.constant_(layer.bias, bias_const)
    return layer


def _tanh_mlp(input_dim, dim, output_dim, num_layers, head_std):
    width_in = int(np.prod(input_dim))
    mods = []
    for _ in range(num_layers):
        mods += [layer_init(nn.Linear(width_in, dim)), nn.Tanh()]
        width_in = dim
    mods.append(layer_init(nn.Linear(dim, int(np.prod(output_dim))), head_std))
    return nn.Sequential(*mods)


class discrete_net(nn.Module):
    """Logits over ``output_dim`` actions (src/nets/nets.py:19-29).  ``dropout`` is accepted and
    unused, as upstream."""

    def __init__(self, dim: int, input_dim, output_dim, num_layers: int, dropout: float, action_std=0.01):
        super().__init__()
        self.net = _tanh_mlp(input_dim, dim, output_dim, num_layers, action_std)

    def forward(self, input):
        return self.net(input)


class continuous_net(nn.Module):
    """Gaussian mean head (src/nets/nets.py:31-41)."""

    def __init__(self, dim: int, input_dim, output_dim, num_layers: int, dropout: float, action_std=0.01):
        super().__init__()
        self.net = _tanh_mlp(input_dim, dim, output_dim, num_layers, action_std)

    def forward(self, input):
        return self.net(input)


class critic(nn.Module):
    """State-value head (src/nets/nets.py:43-53)."""

    def __init__(self, dim: int, input_dim, num_layers: int, dropout: float, action_std=1.0):
        super().__init__()
        self.net = _tanh_mlp(input_dim, dim, 1, num_layers, action_std)

    def forward(self, input):
        return self.net(input)
