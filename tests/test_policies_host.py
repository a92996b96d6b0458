"""CPU: the ``policies`` API surface (policy ABC, bulletArmPolicy, ppoBullet) and ppoBullet.update's
intended two-optimizer PPO step, with the oracle standing in for the kernels."""
import collections
import inspect

import numpy as np
import pytest
import torch
from torch import nn

from aur_ppo_amd.policies import bulletArmPolicy, policy, ppoBullet
from tests import oracle_ops

T = collections.namedtuple("T", "state obs action reward done step_left value expert_action log_probs")


class Pi(nn.Module):
    """Minimal actor with the reference's ``sample(x, action=None)`` contract (src/nets/nets.py:86-105)."""

    def __init__(self):
        super().__init__()
        self.f = nn.Sequential(nn.Flatten(), nn.Linear(2 * 8 * 8, 5))
        self.logstd = nn.Parameter(torch.zeros(5))

    def sample(self, x, action=None):
        mean = self.f(x.float())
        dist = torch.distributions.Normal(mean, self.logstd.exp().expand_as(mean))
        if action is None:
            action = dist.rsample()
        return action, dist.log_prob(action).sum(1, keepdim=True), torch.tanh(mean), dist.entropy()


def test_abc_and_signatures():
    with pytest.raises(TypeError):
        policy()
    assert [m for m in ("load_info", "_loadBatchToDevice", "initNet", "update", "act", "save_agent")
            if getattr(policy, m).__isabstractmethod__] == ["load_info", "_loadBatchToDevice", "initNet", "update", "act", "save_agent"]
    sig = inspect.signature(ppoBullet.__init__).parameters
    for k, v in dict(alpha=1e-2, actor_lr=1e-3, critic_lr=1e-3, gamma=0.99, gae=True, num_processes=5, total_steps=10000,
                     update_epochs=10, clip_coeff=0.2, max_grad_norm=0.5, value_coeff=0.5, expert_weight=0.01,
                     entropy_coeff=0.01, gae_lambda=0.95, clip_vloss=False, norm_adv=True, num_minibatches=32,
                     target_kl=0.01).items():
        assert sig[k].default == v, k
    assert list(inspect.signature(ppoBullet.update).parameters) == ["self", "data", "next_obs", "next_done", "dists"]
    for m in ("initNet", "_loadBatchToDevice", "load_info", "_loadLossCalcDict", "get_buffer_values", "run_gae",
              "normal_advantage", "advantages", "compute_loss_pi", "compute_loss_v", "update", "act", "pretrain_update",
              "save_agent", "decodeActions", "getActionFromPlan"):
        assert callable(getattr(ppoBullet, m)), m


def test_decode_actions_roundtrip_and_pixel_scaling():
    p = ppoBullet(num_processes=2, total_steps=4, num_minibatches=2, ops=oracle_ops)
    plan = torch.tensor([[1.0, 0.01, -0.02, 0.03, 0.1], [0.0, -0.2, 0.2, 0.0, -1.0]])
    unscaled, scaled = p.getActionFromPlan(plan)
    assert unscaled.abs().max() <= 1.0
    np.testing.assert_allclose(scaled[0].numpy(), plan[0].numpy(), atol=1e-6)       # inside the ranges: identity
    np.testing.assert_allclose(scaled[1].numpy(), [0.0, -0.05, 0.05, 0.0, -np.pi / 8], atol=1e-6)   # clamped
    batch = [T(1, np.full((8, 8), 255, np.float32), np.zeros(5, np.float32), np.float32(1.0), 0, 5, np.float32(0.3),
               np.zeros(5, np.float32), np.float32(-1.0))] * 4
    p._loadBatchToDevice(batch, device="cpu")
    assert p.loss_calc_dict["batch_size"] == 2 and p.loss_calc_dict["obs"].shape == (4, 1, 8, 8)
    np.testing.assert_allclose(float(p.loss_calc_dict["obs"].max()), 0.4, rtol=1e-6)   # /255*0.4
    assert p._loadLossCalcDict()[2].shape == (4, 2, 8, 8)                           # state tiled as channel 2


def test_update_moves_both_networks_and_stops_on_kl():
    torch.manual_seed(0)
    p = ppoBullet(num_processes=2, total_steps=6, num_minibatches=3, update_epochs=3, target_kl=1e9, ops=oracle_ops,
                  clip_vloss=True)
    p.device = torch.device("cpu")
    pi, critic = Pi(), nn.Sequential(nn.Flatten(), nn.Linear(2 * 8 * 8, 1))
    p.initNet(pi, critic, "cnn")
    rs = np.random.RandomState(0)
    data = [T(int(rs.randint(2)), (rs.rand(8, 8) * 255).astype(np.float32), rs.randn(5).astype(np.float32),
              np.float32(rs.rand()), int(rs.rand() < 0.2), 5, np.float32(rs.randn()), rs.randn(5).astype(np.float32),
              np.float32(-5 + rs.randn())) for _ in range(12)]
    w0, c0 = pi.f[1].weight.clone(), critic[1].weight.clone()
    p.update(data, torch.rand(2, 2, 8, 8), torch.zeros(2))
    assert not torch.equal(w0, pi.f[1].weight) and not torch.equal(c0, critic[1].weight)
    assert p.last_scalars.shape == (9, 3) and np.isfinite(p.last_scalars).all() and p.loss_calc_dict == {}
    p.target_kl = -1.0                      # stops after the first epoch
    p.update(data, torch.rand(2, 2, 8, 8), torch.zeros(2))
    assert p.last_scalars.shape == (3, 3)
