"""``actor_critic`` with the reference's API (src/models/actor_critic.py:8-51):
``actor_critic(state_dim, action_dim, hidden_dim, num_layers, dropout, continuous)``,
``.value(state) -> (B,)``, ``.evaluate(state, action=None) -> (action, log_prob (B,), entropy (B,),
value (B,1))``; sub-modules ``actor`` / ``critic`` and the state-independent ``actor_logstd (1, A)``.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn

from .nets import continuous_net, critic, discrete_net

_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


class actor_critic(nn.Module):
    def __init__(self, state_dim: int, action_dim, hidden_dim: int, num_layers: int, dropout, continuous: bool) -> None:
        super().__init__()
        self.state_dim = state_dim
        self.action_dim = action_dim
        self.hidden_dim = hidden_dim
        self.continuous = continuous
        self.num_layers = num_layers
        self.dropout = dropout
        # construction order (actor, critic, logstd) fixes the RNG draws of the orthogonal init
        if continuous:
            self.actor = continuous_net(hidden_dim, state_dim, action_dim, num_layers, dropout)
            self.critic = critic(hidden_dim, state_dim, num_layers, dropout)
            self.actor_logstd = nn.Parameter(torch.zeros(1, int(np.prod(action_dim))))
        else:
            self.actor = discrete_net(hidden_dim, state_dim, action_dim, num_layers, dropout)
            self.critic = critic(hidden_dim, state_dim, num_layers, dropout)

    def forward(self):
        pass

    def value(self, state):
        return self.critic(state).flatten()

    def evaluate(self, state, action=None):
        """Normal(mean, exp(logstd)) / Categorical(logits) log-prob and entropy, written out in
        closed form (same values as torch.distributions, src/models/actor_critic.py:34-51, without
        building distribution objects or validating arguments on the hot path)."""
        if self.continuous:
            mean = self.actor(state)
            logstd = self.actor_logstd.expand_as(mean)
            std = torch.exp(logstd)
            if action is None:
                action = torch.normal(mean, std)           # == Normal(mean, std).sample()
            z = (action - mean)
            log_prob = (-(z * z) / (2 * std * std) - logstd - _HALF_LOG_2PI).sum(1)
            entropy = (0.5 + _HALF_LOG_2PI + logstd).sum(1)
        else:
            logits = self.actor(state)
            logp_all = logits - logits.logsumexp(dim=-1, keepdim=True)
            if action is None:
                action = torch.multinomial(logp_all.exp(), 1).squeeze(-1)   # == Categorical.sample()
            log_prob = logp_all.gather(-1, action.long().unsqueeze(-1)).squeeze(-1)
            p = logp_all.exp()
            entropy = -(p * logp_all).sum(-1)
        return action, log_prob, entropy, self.critic(state)
