"""Tanh MLP policy / value networks with the reference's constructor signatures and state-dict
layout (src/nets/nets.py:14-53): ``<net>.net.{0,2,4,...}.{weight,bias}``, orthogonal init with
gain sqrt(2) on hidden layers, 0.01 on the actor head, 1.0 on the critic head, zero biases.
These run as stock PyTorch-ROCm GEMMs (hipBLASLt); they are host code, not custom kernels."""
from __future__ import annotations

import math
import os

import numpy as np
import torch
from torch import nn


def _bf16x3_linear(x, in_features, out_features):
    """Route this product through csrc/conv.hip::k_linear?  Off by default: measured through ``bench.py --hidden-dim 256 --num-layers 2``
    (the per-op path these layers run on) the update took 40.5 ms with it against 39.1 ms on the library's GEMMs -- that path is
    bound by its ~100 launches and element-wise passes per minibatch, and the separate bias pass this route adds costs what the
    faster product saves.  ``AURPPO_LINEAR_BF16X3=1`` switches it on (tests/test_conv_gpu.py holds it to the fp64 product)."""
    if not x.is_cuda or os.environ.get("AURPPO_LINEAR_BF16X3") != "1":
        return False
    from . import hip_ops as H
    return H.linear_ok(x, in_features, out_features)


class _LinearSplitK(torch.autograd.Function):
    """``y = x W^T + b`` whose weight gradient is formed with an explicit split over the batch.

    Measured on MI355X (profiles/r01): for the policy's shapes, x (131072, 64), the library's
    ``dW = dy^T x`` GEMM reduces over K = 131072 inside a single 64x64 output tile and takes
    ~365 us (2 % of the fp32 MFMA rate), while forward and dX take ~22 us.  Splitting the batch into
    S slabs (one small batched GEMM per slab, then a sum over slabs) gives the same fp32 result up
    to summation order in ~26 us.  Forward and dX are the stock hipBLASLt calls."""

    SLABS = 128
    MIN_ROWS = 4096

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        if _bf16x3_linear(x, weight.shape[1], weight.shape[0]):
            # layers the fused steps do not cover (hidden_dim > 128): the product on bf16 MFMAs over three-way splits
            # (csrc/conv.hip::k_linear, fp32-equivalent) instead of the library's fp32 GEMM at 0.35 of the fp32 MFMA peak
            from . import hip_ops as H
            return H.linear_nobias(x.detach(), weight.detach(), 0) + bias
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            if _bf16x3_linear(gy, weight.shape[0], weight.shape[1]):
                from . import hip_ops as H
                gx = H.linear_nobias(gy, weight.detach(), 1)
            else:
                gx = gy @ weight
        m = x.shape[0]
        s = _LinearSplitK.SLABS
        if m % s == 0 and m >= _LinearSplitK.MIN_ROWS:
            gw = torch.bmm(gy.view(s, m // s, -1).transpose(1, 2), x.view(s, m // s, -1)).sum(0)
        else:
            gw = gy.t() @ x
        return gx, gw, gy.sum(0)


class _LinearTanhFn(torch.autograd.Function):
    """``tanh(x W^T + b)`` as ONE kernel forward (csrc/conv.hip::k_linear with bias + tanh in its epilogue) -- a hidden layer of
    src/nets/nets.py:21-27 for the shapes the fused K7 / K7w steps do not cover; backward: tanh' from the saved output, the input
    gradient on the same kernel (mode 1), the weight gradient on k_linear_wgrad where both of its dimensions fill a 128 x 128 tile,
    else with the split over the batch of ``_LinearSplitK``."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from . import hip_ops as H
        h = H.linear_bias_act(x.detach(), weight.detach(), bias.detach(), 1)
        ctx.save_for_backward(x, weight, h)
        return h

    @staticmethod
    def backward(ctx, gh):
        from . import hip_ops as H
        x, weight, h = ctx.saved_tensors
        gz = torch.ops.aten.tanh_backward(gh.contiguous(), h)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = H.linear_nobias(gz, weight.detach(), 1) if H.linear_ok(gz, weight.shape[0], weight.shape[1]) else gz @ weight
        m, s = x.shape[0], _LinearSplitK.SLABS
        if min(gz.shape[1], x.shape[1]) >= 128 and H.linear_wgrad_ok(gz, x):
            gw = H.linear_wgrad(gz, x)       # k_linear_wgrad: 142 us at 256 x 256 x 131 072 where the split-batch bmm + sum takes 202
        elif m % s == 0 and m >= _LinearSplitK.MIN_ROWS:
            gw = torch.bmm(gz.view(s, m // s, -1).transpose(1, 2), x.view(s, m // s, -1)).sum(0)
        else:
            gw = gz.t() @ x
        return gx, gw, gz.sum(0)


class _TanhMLP(nn.Sequential):
    """``nn.Sequential`` of (Linear, Tanh)* + head with the reference's state-dict keys; on the GPU a (Linear, Tanh) pair whose
    shape k_linear takes (hip_ops.linear_ok: inner width a multiple of 16, outer a multiple of 32, >= 32 768 rows) runs as one
    fused kernel: by default for layers wider than 128 -- the shapes the fused K7 / K7w steps do not cover (bench.py --hidden-dim
    256: 38.3 -> 36.8 ms per update with two layers, 58.7 -> 52.9 ms with three) -- and for every such pair with
    ``AURPPO_LINEAR_BF16X3=1``; ``=0`` keeps torch's modules."""

    def forward(self, x):
        mods = list(self)
        i = 0
        env = os.environ.get("AURPPO_LINEAR_BF16X3", "")
        fuse = x.is_cuda and torch.is_grad_enabled() and env != "0"
        if fuse:
            from . import hip_ops as H
        while i < len(mods):
            m = mods[i]
            if (fuse and isinstance(m, nn.Linear) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.Tanh) and x.dim() == 2
                    and (env == "1" or max(m.in_features, m.out_features) > 128)
                    and H.linear_ok(x, m.in_features, m.out_features)):
                x = _LinearTanhFn.apply(x, m.weight, m.bias)
                i += 2
            else:
                x = m(x)
                i += 1
        return x


class _Linear(nn.Linear):
    """nn.Linear (same parameters / state-dict keys) with the split-batch weight gradient."""

    def forward(self, input):
        if input.dim() == 2 and input.is_cuda and torch.is_grad_enabled() and input.shape[0] >= _LinearSplitK.MIN_ROWS:
            return _LinearSplitK.apply(input, self.weight, self.bias)
        return super().forward(input)


def layer_init(layer, std=math.sqrt(2), bias_const=0.0):
    nn.init.orthogonal_(layer.weight, std)
    nn.init.constant_(layer.bias, bias_const)
    return layer


def _tanh_mlp(input_dim, dim, output_dim, num_layers, head_std):
    width_in = int(np.prod(input_dim))
    mods = []
    for _ in range(num_layers):
        mods += [layer_init(_Linear(width_in, dim)), nn.Tanh()]
        width_in = dim
    mods.append(layer_init(_Linear(dim, int(np.prod(output_dim))), head_std))
    return _TanhMLP(*mods)


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
