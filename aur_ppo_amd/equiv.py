"""C4-equivariant encoder / actor / critic with the module layout of the reference's e2cnn networks
(src/nets/equiv.py:12-91 ``EquivariantEncoder128`` + ``EquivariantActor``, :127-157 ``EquivariantCritic``): the same
blocks, the same field types per layer (``n_out//8, //4, //2, 1, 2, 1, 1`` regular C4 fields; actor head = one irrep(1)
field for (dx, dy) + ``2*action_dim - 2`` trivial fields; critic head = 1x1 regular conv, ReLU, group pooling, 1x1 linear),
the same outputs (``mean, log_std`` / invariant value).

BUILD-DEFINED, parity UNPINNED: upstream's arithmetic lives in e2cnn 0.2.3 (``R2Conv`` expands its filters in a
band-limited steerable basis with its own initialisation), which is neither vendored nor installable here and which no
reference test pins.  This module implements the C4 group convolution directly instead -- for the cyclic group of 90-degree
rotations a 3x3 filter rotates EXACTLY (``torch.rot90``), so a regular-representation group convolution is an ordinary
``conv2d`` with a filter bank built from rotated copies of the free parameters:

    trivial -> regular :  W[(o, g), i]       = rot90^g( psi[o, i] )
    regular -> regular :  W[(o, g), (i, h)]  = rot90^g( psi[o, i, (h - g) mod 4] )

Its function class (all C4-equivariant 3x3 filters) contains e2cnn's; its weights and initialisation are its own, so numbers
cannot be compared with upstream's -- what IS tested is the defining property: rotating the observation by 90 degrees
rotates (dx, dy) by 90 degrees and leaves the other action means, every log-std and the value unchanged
(tests/test_equiv.py, CPU and GPU).  The convolutions are MIOpen's; conv + ReLU + max-pool blocks use K9 on the GPU.

An 84x84 variant (BASELINE config 5) keeps every spatial op rotation-symmetric: 84 -> 42 -> 21 -(pool 3)-> 7 -> 5 -> 3 -> 1.
"""
from __future__ import annotations

import math
import os

import torch
from torch import nn
from torch.nn import functional as F

LOG_SIG_MAX = 2
LOG_SIG_MIN = -20


class C4Conv(nn.Module):
    """Group convolution for C4 = {rotations by k*90 degrees}.  ``in_type`` / ``out_type``: "trivial" (one channel per
    field) or "regular" (four channels per field, channel g = response of the filter rotated g times); a regular output
    field's four channels share one bias.  Channel order is field-major: channel = field * 4 + g."""

    def __init__(self, in_fields, out_fields, in_type="regular", out_type="regular", kernel_size=3, padding=0):
        super().__init__()
        assert in_type in ("trivial", "regular") and out_type in ("trivial", "regular")
        assert not (in_type == "regular" and out_type == "trivial"), "use GroupPool + a trivial->trivial C4Conv"
        self.in_fields, self.out_fields, self.in_type, self.out_type = in_fields, out_fields, in_type, out_type
        self.kernel_size, self.padding = kernel_size, padding
        k = kernel_size
        gi = 4 if in_type == "regular" else 1
        shape = (out_fields, in_fields, gi, k, k)
        self.weight = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.zeros(out_fields))
        nn.init.normal_(self.weight, 0.0, math.sqrt(2.0 / (in_fields * gi * k * k)))      # He: fan-in of one output channel

    def expanded_weight(self):
        w = self.weight
        O, I, gi, k, _ = w.shape
        if self.out_type == "trivial":            # trivial -> trivial: an ordinary filter must itself be invariant:
            sym = sum(torch.rot90(w[:, :, 0], g, (2, 3)) for g in range(4)) / 4.0
            return sym                            # (O, I, k, k)
        rows = []
        for g in range(4):
            r = torch.rot90(w, g, (3, 4))                                   # rotate every filter g times
            if gi == 4:
                r = torch.roll(r, shifts=g, dims=2)                         # input channel h reads psi[(h - g) mod 4]
            rows.append(r.reshape(O, I * gi, k, k))
        return torch.stack(rows, dim=1).reshape(O * 4, I * gi, k, k)        # out channel = o * 4 + g

    def expanded_bias(self):
        return self.bias.repeat_interleave(4) if self.out_type == "regular" else self.bias

    fused_conv = os.environ.get("AURPPO_NO_K11") != "1"      # K11 (csrc/conv.hip) for the expanded filter bank when its shape rule holds

    def forward(self, x, with_bias=True):
        w = self.expanded_weight()
        if self.fused_conv and x.is_cuda and self.kernel_size == 3:
            from . import hip_ops as H
            if H.conv3x3_ok(x, w.shape[1], w.shape[0], self.padding):
                # the expanded C4 filter bank is a dense 3x3 convolution over 4 x fields channels: the contraction north_star
                # reserves the matrix cores for.  The gradient reaches the free parameters through expanded_weight's autograd.
                z = H.conv3x3(x, w, self.padding)
                return z + self.expanded_bias().reshape(1, -1, 1, 1) if with_bias else z
        return F.conv2d(x, w, self.expanded_bias() if with_bias else None, padding=self.padding)


class GroupPool(nn.Module):
    """Max over the four group channels of every regular field -> one invariant (trivial) channel per field
    (e2cnn ``GroupPooling``, src/nets/equiv.py:143)."""

    def forward(self, x):
        B, C = x.shape[:2]
        return x.reshape(B, C // 4, 4, *x.shape[2:]).amax(dim=2)


class _Block(nn.Module):
    """C4Conv + ReLU (+ max-pool): on the GPU with a 2x2 pool the bias / ReLU / pool tail is K9."""

    def __init__(self, conv, pool=0):
        super().__init__()
        self.conv, self.pool = conv, pool

    def forward(self, x):
        if self.pool == 2 and x.is_cuda and getattr(self, "fused_pool", True):
            from . import hip_ops as H
            return H.bias_relu_pool2(self.conv(x, with_bias=False), self.conv.expanded_bias())
        y = F.relu(self.conv(x))
        return F.max_pool2d(y, self.pool) if self.pool else y


class EquivariantEncoder(nn.Module):
    """Field layout of ``EquivariantEncoder128`` (src/nets/equiv.py:12-61) for 128x128 inputs; a rotation-symmetric
    build-defined layout for 84x84.  Output: (B, n_out * 4, 1, 1) -- ``n_out`` regular fields."""

    def __init__(self, obs_channel=2, n_out=128, obs_size=128):
        super().__init__()
        n = n_out
        if obs_size == 128:
            spec = [(obs_channel, n // 8, "trivial", 1, 2), (n // 8, n // 4, "regular", 1, 2), (n // 4, n // 2, "regular", 1, 2),
                    (n // 2, n, "regular", 1, 2), (n, 2 * n, "regular", 1, 0), (2 * n, n, "regular", 0, 2), (n, n, "regular", 0, 0)]
        elif obs_size == 84:
            spec = [(obs_channel, n // 8, "trivial", 1, 2), (n // 8, n // 4, "regular", 1, 2), (n // 4, n // 2, "regular", 1, 3),
                    (n // 2, n, "regular", 0, 0), (n, n, "regular", 0, 0), (n, n, "regular", 0, 0)]
        else:
            raise ValueError(f"EquivariantEncoder: observations must be 128x128 or 84x84, got {obs_size}")
        self.conv = nn.Sequential(*[_Block(C4Conv(i, o, t, "regular", 3, pad), pool) for (i, o, t, pad, pool) in spec])

    def forward(self, x, state=None):
        """``x``: the observation with the tiled gripper state as its last channel (upstream's input), or -- with ``state``
        given -- the image channels alone: on the GPU the first block then runs as K10 (the state is a constant plane, a
        trivial input field, so nothing about the symmetry changes)."""
        if state is None:
            return self.conv(x)
        first = self.conv[0]
        if (x.is_cuda and first.pool == 2 and getattr(first, "fused_pool", True) and x.shape[1] <= 3
                and (first.conv.out_fields * 4) % 16 == 0):
            from . import hip_ops as H
            y = H.first_block(x, state, first.conv.expanded_weight(), first.conv.expanded_bias())
            return self.conv[1:](y)
        cat = torch.cat([x, state.reshape(-1, 1, 1, 1).to(x.dtype).expand(x.shape[0], 1, x.shape[2], x.shape[3])], dim=1)
        return self.conv(cat)


class EquivariantActor(nn.Module):
    """src/nets/equiv.py:64-91: encoder, then a 1x1 map from the regular fields to one irrep(1) field -- the (dx, dy)
    mean, which rotates WITH the observation -- and ``2*action_dim - 2`` trivial fields (the invariant means p, dz,
    dtheta and all ``action_dim`` log-stds)."""

    def __init__(self, obs_shape=(2, 128, 128), action_dim=5, n_hidden=128):
        super().__init__()
        self.obs_channel, self.action_dim, self.n_hidden = obs_shape[0], action_dim, n_hidden
        self.enc = EquivariantEncoder(obs_shape[0], n_hidden, obs_shape[1])
        s = math.sqrt(1.0 / (4 * n_hidden))
        self.w_vec = nn.Parameter(torch.randn(n_hidden, 2) * s)               # u_i: the field's contribution to (dx, dy) at g = 0
        self.w_inv = nn.Parameter(torch.randn(2 * action_dim - 2, n_hidden) * s)
        self.b_inv = nn.Parameter(torch.zeros(2 * action_dim - 2))            # (an irrep(1) output admits no bias)

    def forward(self, obs, state=None):
        B = obs.shape[0]
        f = self.enc(obs, state).reshape(B, self.n_hidden, 4)                 # (B, field, g)
        # R(g) u for the four rotations: (x, y) -> (-y, x) per quarter turn
        u = self.w_vec
        rot = torch.stack([u, torch.stack([-u[:, 1], u[:, 0]], 1), -u, torch.stack([u[:, 1], -u[:, 0]], 1)], dim=1)   # (field, g, 2)
        dxy = torch.einsum("bfg,fgc->bc", f, rot)
        inv = f.sum(2) @ self.w_inv.t() + self.b_inv                          # constant over g: invariant
        A = self.action_dim
        inv_act = inv[:, :A - 2]
        mean = torch.cat((inv_act[:, 0:1], dxy, inv_act[:, 1:]), dim=1)       # (p, dx, dy, dz, dtheta)
        log_std = torch.clamp(inv[:, A - 2:], min=LOG_SIG_MIN, max=LOG_SIG_MAX)
        return mean, log_std


class _GeoValue(torch.Tensor):
    """The reference's equivariant critic returns an e2cnn ``GeometricTensor`` and its callers read ``.tensor``
    (src/robot_ppo.py:165,264,374): this tensor answers to both."""

    @property
    def tensor(self):
        return self.as_subclass(torch.Tensor)


class EquivariantCritic(nn.Module):
    """src/nets/equiv.py:127-157: encoder, 1x1 regular conv, ReLU, group pooling, 1x1 linear -> one invariant value."""

    def __init__(self, obs_shape=(2, 128, 128), n_hidden=128):
        super().__init__()
        self.n_hidden = n_hidden
        self.img_conv = EquivariantEncoder(obs_shape[0], n_hidden, obs_shape[1])
        self.critic = nn.Sequential(_Block(C4Conv(n_hidden, n_hidden, "regular", "regular", 1, 0)), GroupPool(),
                                    nn.Conv2d(n_hidden, 1, kernel_size=1))

    def forward(self, obs, state=None):
        out = self.critic(self.img_conv(obs, state))
        return out.as_subclass(_GeoValue)
