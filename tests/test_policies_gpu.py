"""GPU: the ``policies`` / ``trainer`` caller family and the resume path on the real HIP kernels."""
import collections

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


class Pi(nn.Module):
    """Small conv actor with the reference's ``sample(x, action=None)`` contract (src/nets/nets.py:86-105: tanh-squashed
    Gaussian, log-prob corrected for the squashing, per-element entropy)."""

    def __init__(self):
        super().__init__()
        self.f = nn.Sequential(nn.Conv2d(2, 4, 3, padding=1), nn.ReLU(), nn.Flatten(), nn.Linear(4 * 8 * 8, 5))
        self.logstd = nn.Parameter(torch.full((5,), -0.5))

    def sample(self, x, action=None):
        mean = self.f(x.float())
        dist = torch.distributions.Normal(mean, self.logstd.exp().expand_as(mean))
        if action is None:
            action = dist.rsample()
        a = torch.tanh(action)
        lp = (dist.log_prob(action) - torch.log((1 - a.pow(2)) + 1e-6)).sum(1, keepdim=True)
        return a, lp, torch.tanh(mean), dist.entropy()


def _critic():
    return nn.Sequential(nn.Conv2d(2, 4, 3, padding=1), nn.ReLU(), nn.Flatten(), nn.Linear(4 * 8 * 8, 1))


@pytest.mark.parametrize("clip_vloss,gae", [(True, True), (False, True), (True, False)])
def test_ppo_bullet_update_matches_oracle_restatement(clip_vloss, gae):
    """policies.ppoBullet.update on K1 + K4/K5 vs oracle.reference_ppobullet_update (src/policies/ppoBullet.py:123-298 as
    intended): same rollout, same nets, same two Adam optimizers -- per-step losses, KL and final weights."""
    import copy
    from aur_ppo_amd.policies import ppoBullet
    from oracle import ppo_oracle as O
    T, N, A = 6, 4, 5
    B = T * N
    torch.manual_seed(3)
    pi, critic = Pi(), _critic()
    pi_c, critic_c = copy.deepcopy(pi), copy.deepcopy(critic)
    g = torch.Generator().manual_seed(8)
    pixels = torch.rand(B, 1, 8, 8, generator=g) * 255                  # raw pixels; the policy scales /255*0.4
    states = (torch.rand(B, generator=g) < 0.5).float()
    with torch.no_grad():     # actions the policy itself could have taken (pre-tanh), their log-probs, critic values
        obs_t = torch.cat([pixels / 255 * 0.4, states.reshape(-1, 1, 1, 1).repeat(1, 1, 8, 8)], 1)
        raw = pi_c.f(obs_t) + 0.6 * torch.randn(B, A, generator=g)
        _, lp, _, _ = pi_c.sample(obs_t, raw)
        val = critic_c(obs_t).reshape(-1)
    dense = dict(state=states, obs=pixels, action=raw, reward=torch.rand(B, generator=g),
                 done=(torch.rand(B, generator=g) < 0.15).float(), step_left=torch.full((B,), 100.0),
                 value=val + 0.1 * torch.randn(B, generator=g), expert_action=torch.tanh(torch.randn(B, A, generator=g)),
                 log_probs=lp.reshape(-1) + 0.05 * torch.randn(B, generator=g))
    next_obs = torch.cat([torch.rand(N, 1, 8, 8, generator=g) * 0.4, (torch.rand(N, generator=g) < 0.5).float()
                          .reshape(-1, 1, 1, 1).repeat(1, 1, 8, 8)], 1)
    next_done = (torch.rand(N, generator=g) < 0.2).float()
    hp = dict(gamma=0.99, gae_lambda=0.95, gae=gae, clip_coeff=0.2, entropy_coeff=0.01, value_coeff=0.5, expert_weight=0.01,
              norm_adv=True, clip_vloss=clip_vloss, num_update_epochs=3, minibatch_size=8, target_kl=1e9)
    # HIP path
    agent = ppoBullet(num_processes=N, total_steps=T, num_minibatches=3, update_epochs=3, target_kl=1e9, gae=gae,
                      clip_vloss=clip_vloss)
    assert agent.device.type == "cuda"
    agent.initNet(pi.cuda(), critic.cuda(), "cnn")
    agent.update({k: v.cuda() for k, v in dense.items()}, next_obs.cuda(), next_done.cuda())
    torch.cuda.synchronize()
    # oracle
    pi_opt = torch.optim.Adam([{"params": pi_c.parameters(), "lr": 1e-3}])
    v_opt = torch.optim.Adam(critic_c.parameters(), lr=1e-3)
    batch = dict(states=states, obs=pixels / 255 * 0.4, actions=raw, rewards=dense["reward"], dones=dense["done"],
                 values=dense["value"], log_probs=dense["log_probs"], expert=dense["expert_action"])
    rows, ret_o, adv_o = O.reference_ppobullet_update(pi_c, critic_c, pi_opt, v_opt, batch, next_obs, next_done, hp, N)
    got = agent.last_scalars
    assert got.shape == rows.shape == (9, 3)
    np.testing.assert_allclose(got, rows, rtol=2e-4, atol=2e-5)
    for (k, a), (_, b) in zip(list(agent.pi.state_dict().items()) + list(agent.critic.state_dict().items()),
                              list(pi_c.state_dict().items()) + list(critic_c.state_dict().items())):
        # nine Adam steps at lr 1e-3: a near-zero gradient element may flip a whole step
        d = np.abs(a.cpu().numpy() - b.numpy())
        assert d.max() <= 2e-3 and np.mean(d > 1e-4) < 0.05, (k, d.max(), np.mean(d > 1e-4))


def test_ppo_bullet_trainer_runs_on_the_gpu():
    from aur_ppo_amd.policies import ppoBullet
    from aur_ppo_amd.trainer import ppoBulletTrainer
    torch.manual_seed(0)
    np.random.seed(0)
    agent = ppoBullet(num_processes=4, total_steps=8, num_minibatches=4, update_epochs=2, target_kl=1e9, clip_vloss=True)
    tr = ppoBulletTrainer(agent, anneal_lr=True, total_time_steps=4 * 8 * 2, num_env_steps=8, num_processes=4,
                          pretrain_episodes=2, num_eval_episodes=1)
    pi, critic = Pi().cuda(), _critic().cuda()
    w0 = pi.f[3].weight.clone()
    last = tr.run(None, {"obs_size": 8}, {}, "Synthetic-arm", pi, critic, "cnn", log=False)
    assert tr.replay_buffer.obs.is_cuda and last.shape == (8, 3) and np.isfinite(last).all()
    assert not torch.equal(w0, pi.f[3].weight)


def test_checkpoint_resume_on_the_gpu(tmp_path):
    """Same contract as tests/test_trainer_host.py::test_checkpoint_resume_continues_the_same_run, on the HIP path: flat
    Adam moments, device step counter, device shuffle generator and the permutations drawn ahead all round-trip."""
    from aur_ppo_amd.ppo import ppo
    from tests.test_parity_fullsize import _hp
    from tests.util import synth_rollout
    T, N, Dm, A = 16, 64, 16, 3
    d = synth_rollout(T, N, Dm, A, seed=3)

    def drive(agent, u0, u1):
        for u in range(u0, u1):
            agent.set_lr((1 - u / 4) * 3e-4)
            agent.buffer.rewards.copy_(torch.from_numpy(d["rewards"]) + 0.1 * u)
            ret, adv = agent.advantages(torch.from_numpy(d["next_obs"]).cuda(), torch.from_numpy(d["next_done"]).cuda())
            agent.update(ret, adv)

    def fresh():
        torch.manual_seed(5)
        a = ppo(_hp(N, T, Dm, A, hip_graph=False, num_update_epochs=2))
        for k in ("states", "actions", "log_probs", "terminals", "values"):
            getattr(a.buffer, k).copy_(torch.from_numpy(d[k]))
        a.seed_all(1)
        return a

    a = fresh()
    drive(a, 0, 4)
    b = fresh()
    drive(b, 0, 2)
    path = str(tmp_path / "ck.pt")
    b.save_checkpoint(path, update=2)
    c = fresh()
    with torch.no_grad():
        c.bucket.flat_param[:c.bucket.numel].add_(1.0)
    assert c.load_checkpoint(path) == 2
    assert float(c._adam_t) == 16.0
    drive(c, 2, 4)
    torch.cuda.synchronize()
    assert torch.equal(c.bucket.flat_param, a.bucket.flat_param)
    assert torch.equal(c._adam_m, a._adam_m) and torch.equal(c._adam_v, a._adam_v)
    assert torch.equal(c._last_perms, a._last_perms)
