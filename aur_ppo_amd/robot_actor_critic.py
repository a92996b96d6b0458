"""``robot_actor_critic`` with the reference's API (src/models/robot_actor_critic.py:19-157):
``robot_actor_critic(device, equivariant, dx=.02, dy=.02, dz=.02, dr=pi/8, n_a=5, tau=.001)``,
``.evaluate(state, obs, action=None) -> (actions, unscaled_actions, log_prob (B,), entropy (B,), value)``,
``.value(state, obs)``, ``.decodeActions(*cols)``, ``.getActionFromPlan(plan)``, ``.test_action``.

``equivariant=True`` selects upstream's C4-equivariant actor / critic layout (src/nets/equiv.py).  Upstream builds it on
e2cnn 0.2.3, which is neither vendored nor installed and which no reference test pins: aur_ppo_amd/equiv.py implements the
same field layout directly (exact 90-degree filter rotations) -- BUILD-DEFINED weights, parity unpinned, equivariance
property-tested."""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn

from .base_cnns import base_actor, base_critic, base_encoder, weights_init

_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


class robot_actor_critic(nn.Module):
    def __init__(self, device, equivariant: bool, dx=0.02, dy=0.02, dz=0.02, dr=np.pi / 8, n_a=5, tau=0.001,
                 obs_shape=(1, 128, 128), n_hidden=128) -> None:
        """``obs_shape`` (extra, last): (channels, H, W) of the image observation -- (1, 128, 128) upstream; (3, 84, 84)
        selects the build-defined encoder of base_cnns.base_encoder.  The gripper state adds one input plane."""
        super().__init__()
        self.obs_shape = tuple(int(x) for x in obs_shape)
        net_in = (self.obs_shape[0] + 1,) + self.obs_shape[1:]
        self.p_range = torch.tensor([0, 1])
        self.dtheta_range = torch.tensor([-dr, dr])
        self.dx_range = torch.tensor([-dx, dx])
        self.dy_range = torch.tensor([-dy, dy])
        self.dz_range = torch.tensor([-dz, dz])
        self.n_a = n_a
        self.device = device
        self.memory_format = None      # torch.channels_last: NHWC activations behind the first convolution (trainer option)
        self.equivariant = equivariant
        if equivariant:
            # src/models/robot_actor_critic.py:33-35: only an actor (mean AND log-std from the network) and a critic
            from .equiv import EquivariantActor, EquivariantCritic
            self.actor = EquivariantActor(obs_shape=net_in, action_dim=n_a, n_hidden=n_hidden)
            self.critic = EquivariantCritic(obs_shape=net_in, n_hidden=n_hidden)
            return
        self.network = base_encoder(obs_shape=net_in, out_dim=128)   # unused upstream too; kept for state dicts
        self.actor = base_actor(obs_shape=net_in)
        self.actor.apply(weights_init)
        self.actor_logstd = nn.Parameter(torch.zeros(1, 5))
        self.critic = base_critic(obs_shape=net_in)
        self.critic.apply(weights_init)

    def forward(self, act):
        pass

    @staticmethod
    def _cat(state, obs):
        return torch.cat([obs, state.reshape(state.size(0), 1, 1, 1).to(obs.dtype).repeat(1, 1, obs.shape[2], obs.shape[3])], dim=1)

    def value(self, state, obs):
        if self.equivariant:          # the tiled state is a constant plane: a trivial (rotation-invariant) input field
            return self.critic(obs.to(self.device), state.to(self.device))
        # upstream tiles the gripper state to a plane and concatenates it (:58-59); folded into conv 1 here
        return self.critic(obs.to(self.device), state.to(self.device), self.memory_format)

    @staticmethod
    def _scale(u, rng):
        return 0.5 * (u + 1) * (rng[1] - rng[0]) + rng[0]

    def decodeActions(self, *args):
        cols = [args[0], args[1], args[2], args[3]]
        rngs = [self.p_range, self.dx_range, self.dy_range, self.dz_range]
        if self.n_a == 5:
            cols.append(args[4])
            rngs.append(self.dtheta_range)
        actions = torch.stack([self._scale(u, r.to(u.device)) for u, r in zip(cols, rngs)], dim=1)
        return torch.stack(cols, dim=1), actions

    def getActionFromPlan(self, plan):
        def unscale(a, rng):
            return 2 * (a - rng[0]) / (rng[1] - rng[0]) - 1
        rngs = [self.p_range, self.dx_range, self.dy_range, self.dz_range] + ([self.dtheta_range] if self.n_a == 5 else [])
        cols = []
        for i, r in enumerate(rngs):
            r = r.to(plan.device)
            cols.append(unscale(plan[:, i].clamp(r[0], r[1]), r))
        return self.decodeActions(*cols)

    def evaluate(self, state, obs, action=None):
        state, obs = state.to(self.device), obs.to(self.device)
        if self.equivariant:          # src/models/robot_actor_critic.py:109-110 (the tile + concat happens inside the nets)
            mean, logstd = self.actor(obs, state)
        else:
            mean = self.actor(obs, state, self.memory_format)
            logstd = self.actor_logstd.expand_as(mean)
        std = torch.exp(logstd)
        if action is None:
            action = mean + std * torch.randn_like(mean)            # == Normal(mean, std).rsample()
        z = action - mean
        log_prob = (-(z * z) / (2 * std * std) - logstd - _HALF_LOG_2PI).sum(1)
        entropy = (0.5 + _HALF_LOG_2PI + logstd).sum(1)
        unscaled_actions, actions = self.decodeActions(*[action[:, i] for i in range(self.n_a)])
        if self.equivariant:
            return actions, unscaled_actions, log_prob, entropy, self.critic(obs, state)
        return actions, unscaled_actions, log_prob, entropy, self.critic(obs, state, self.memory_format)

    def test_action(self, state, obs):
        if self.equivariant:
            mean = torch.tanh(self.actor(obs.to(self.device), state.to(self.device))[0])
            return self.decodeActions(*[mean[:, i] for i in range(self.n_a)])
        mean = torch.tanh(self.actor(obs.to(self.device), state.to(self.device), self.memory_format))
        return self.decodeActions(*[mean[:, i] for i in range(self.n_a)])
