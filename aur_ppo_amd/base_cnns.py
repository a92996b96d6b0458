"""Plain-CNN encoder / actor / critic of the robot policy with the reference's module layout and
state-dict keys (src/nets/base_cnns.py:12-84): ``base_encoder.conv.{0,3,6,9,12,14,17}``,
``base_actor.{conv,mean_linear}``, ``base_critic.{conv,critic.{0,2}}``; xavier init as upstream.
The convolutions are stock PyTorch-ROCm (MIOpen); on the GPU the element-wise tail of every conv + ReLU + max-pool block
(bias add, ReLU, 2x2 max-pool, and for the first block the tiled gripper-state channel) is K9, one fused HIP pass forward
and one backward (csrc/pool.hip): rocprof put 53 % of robot_ppo.update's GPU time in those memory-bound passes."""
from __future__ import annotations

import os

import torch
from torch import nn


def weights_init(m):
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight, gain=1)
        nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.Conv2d):
        nn.init.xavier_normal_(m.weight.data)


class base_encoder(nn.Module):
    """128x128 -> 64 -> 32 -> 16 -> 8 -> (3x3 valid) 6 -> 3 -> (3x3 valid) 1, channels
    in-16-32-64-128-256-256-out_dim (src/nets/base_cnns.py:20-54).

    BUILD-DEFINED variant for 84x84 observations (BASELINE config 5; upstream has no encoder for that size -- its
    equivariant one asserts 128, src/nets/equiv.py:159-162): 84 -> 42 -> 21 -> 10 -> 5 -> (3x3 valid) 3 -> (3x3 valid) 1,
    channels in-16-32-64-128-256-out_dim -- the same conv3x3 + ReLU + maxpool2 blocks, one pooling stage fewer.  It makes
    no equivariance claim and matches no upstream weights; it exists so the image-observation workload has a policy."""

    def __init__(self, obs_shape=(2, 128, 128), out_dim=1024):
        super().__init__()
        mods, c_in = [], obs_shape[0]
        size = int(obs_shape[1])
        if size not in (128, 84) or int(obs_shape[2]) != size:
            raise ValueError(f"base_encoder: observations must be 128x128 (reference) or 84x84 (build-defined), got {obs_shape}")
        for c_out in (16, 32, 64, 128):
            mods += [nn.Conv2d(c_in, c_out, kernel_size=3, padding=1), nn.ReLU(inplace=True), nn.MaxPool2d(2)]
            c_in = c_out
        if size == 128:
            mods += [nn.Conv2d(128, 256, kernel_size=3, padding=1), nn.ReLU(inplace=True),
                     nn.Conv2d(256, 256, kernel_size=3, padding=0), nn.ReLU(inplace=True), nn.MaxPool2d(2),
                     nn.Conv2d(256, out_dim, kernel_size=3, padding=0), nn.ReLU(inplace=True), nn.Flatten()]
        else:
            mods += [nn.Conv2d(128, 256, kernel_size=3, padding=0), nn.ReLU(inplace=True),
                     nn.Conv2d(256, out_dim, kernel_size=3, padding=0), nn.ReLU(inplace=True), nn.Flatten()]
        self.conv = nn.Sequential(*mods)

    fused_pool = True      # K9 on CUDA tensors (False: the stock torch ops everywhere, for A/B runs)
    fused_conv = os.environ.get("AURPPO_NO_K11") != "1"      # K11 for the hidden 3x3 convolutions on CUDA tensors (False / AURPPO_NO_K11=1: torch / MIOpen, for A/B runs)
    fused_first = True     # K10 for the first block when it runs in split form on CUDA tensors

    def _blocks(self, x, start, scale=None, plane=None, first_weight=None):
        """Run self.conv[start:] on x; on the GPU every (Conv2d, ReLU, MaxPool2d(2)) triple becomes convolution without bias
        + K9.  ``scale`` / ``plane`` / ``first_weight`` belong to the first convolution when it runs in split form."""
        mods = list(self.conv)
        use = self.fused_pool and x.is_cuda
        if use:
            from . import hip_ops as H      # raises without the built library: there is no silent fallback on a GPU box
        i = start
        while i < len(mods):
            m = mods[i]
            triple = (isinstance(m, nn.Conv2d) and i + 2 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                      and isinstance(mods[i + 2], nn.MaxPool2d) and mods[i + 2].kernel_size in (2, (2, 2))
                      and mods[i + 2].stride in (2, (2, 2)) and mods[i + 2].padding in (0, (0, 0)))
            if use and triple:
                w = first_weight if (i == start and first_weight is not None) else m.weight
                if self.fused_conv and w is m.weight and H.conv3x3_supported(x, m):
                    z = H.conv3x3(x, w, m.padding[0])
                else:
                    z = torch.nn.functional.conv2d(x, w, None, stride=m.stride, padding=m.padding)
                x = H.bias_relu_pool2(z, m.bias, scale if i == start else None, plane if i == start else None)
                i += 3
            elif use and self.fused_conv and isinstance(m, nn.Conv2d) and H.conv3x3_supported(x, m):
                x = H.conv3x3(x, m.weight, m.padding[0]) + m.bias.reshape(1, -1, 1, 1)
                i += 1
            else:
                x = m(x)
                i += 1
        return x

    def forward(self, x):
        return self._blocks(x, 0)

    def forward_split(self, obs, state, memory_format=None):
        """Same result as ``forward(cat([obs, state tiled to a plane], 1))`` without materialising the
        concatenated (B, C+1, H, W) tensor: convolution is linear in its input channels, so the tiled
        plane contributes ``state * conv(ones)`` -- one tiny per-call map -- to the first layer."""
        first = self.conv[0]
        c = obs.shape[1]
        mods = self.conv
        if (self.fused_first and self.fused_pool and obs.is_cuda and memory_format is None and c <= 3 and first.out_channels % 16 == 0
                and first.kernel_size == (3, 3) and first.padding == (1, 1) and first.stride == (1, 1)
                and isinstance(mods[1], nn.ReLU) and isinstance(mods[2], nn.MaxPool2d) and mods[2].kernel_size in (2, (2, 2))):
            # K10: convolution of the image channels and of the tiled state, bias, ReLU and the pool in one kernel
            from . import hip_ops as H
            return self._blocks(H.first_block(obs, state, first.weight, first.bias), 3)
        if self.fused_pool and obs.is_cuda and memory_format is None:
            # K9 adds state * plane and the bias while it applies ReLU and the pool: the block's output is written once
            ones = torch.ones((1, 1) + tuple(obs.shape[2:]), device=obs.device, dtype=obs.dtype)
            plane = torch.nn.functional.conv2d(ones, first.weight[:, c:c + 1], None, padding=first.padding)
            return self._blocks(obs, 0, scale=state.reshape(-1).to(obs.dtype), plane=plane, first_weight=first.weight[:, :c])
        y = torch.nn.functional.conv2d(obs, first.weight[:, :c], None, padding=first.padding)
        ones = torch.ones((1, 1) + tuple(obs.shape[2:]), device=obs.device, dtype=obs.dtype)
        plane = torch.nn.functional.conv2d(ones, first.weight[:, c:c + 1], None, padding=first.padding)
        y = y + state.reshape(-1, 1, 1, 1) * plane + first.bias.reshape(1, -1, 1, 1)
        if memory_format is not None:
            y = y.contiguous(memory_format=memory_format)   # the layout the remaining convolutions then keep
            return self.conv[1:](y)
        return self._blocks(y, 1)


class base_critic(nn.Module):
    def __init__(self, obs_shape=(2, 128, 128)):
        super().__init__()
        self.conv = base_encoder(obs_shape=obs_shape, out_dim=128)
        self.critic = nn.Sequential(nn.Linear(128, 128), nn.ReLU(inplace=True), nn.Linear(128, 1))
        self.apply(weights_init)

    def forward(self, obs, state=None, memory_format=None):
        feats = self.conv(obs) if state is None else self.conv.forward_split(obs, state, memory_format)
        return self.critic(feats)


class base_actor(nn.Module):
    def __init__(self, obs_shape=(2, 128, 128), action_dim=5):
        super().__init__()
        self.conv = base_encoder(obs_shape=obs_shape, out_dim=128)
        self.mean_linear = nn.Linear(128, action_dim)
        self.apply(weights_init)

    def forward(self, x, state=None, memory_format=None):
        feats = self.conv(x) if state is None else self.conv.forward_split(x, state, memory_format)
        return self.mean_linear(feats)
