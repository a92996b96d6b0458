"""CPU: the ``trainer`` caller family (policyTrainer ABC, bulletTrainer, ppoBulletTrainer over policies.ppoBullet) and
the resume path, with the oracle standing in for the kernels."""
import inspect

import numpy as np
import pytest
import torch
from torch import nn

from aur_ppo_amd.policies import ppoBullet
from aur_ppo_amd.trainer import DenseTransitionBuffer, bulletTrainer, policyTrainer, ppoBulletTrainer
from tests import oracle_ops
from tests.test_policies_host import Pi


def test_abc_and_constructor_signatures_match_reference():
    with pytest.raises(TypeError):
        policyTrainer()
    for m in ("initialize_env", "pretrain", "evaluate", "step_env", "run"):     # src/trainer/policyTrainer.py:21-53
        assert getattr(policyTrainer, m).__isabstractmethod__, m
    sig = inspect.signature(ppoBulletTrainer.__init__).parameters               # src/trainer/ppoBulletTrainer.py:16-17
    for k, v in dict(anneal_lr=False, anneal_exp=False, total_time_steps=100000, num_env_steps=1024, num_processes=5,
                     pretrain_episodes=5000, num_eval_episodes=100, track=False, run_id=0).items():
        assert sig[k].default == v, k
    assert list(inspect.signature(ppoBulletTrainer.run).parameters)[:8] == [
        "self", "simulator", "env_config", "planner_config", "gym_id", "actor", "critic", "encoder_type"]
    assert list(inspect.signature(bulletTrainer.__init__).parameters)[1:4] == ["total_time_steps", "num_env_steps", "num_processes"]


def test_dense_buffer_is_time_major_and_resets():
    b = DenseTransitionBuffer(3, 2, (1, 4, 4), 5, "cpu")
    for t in range(3):
        b.add_step(state=torch.full((2,), float(t)), obs=torch.full((2, 1, 4, 4), float(t)), action=torch.zeros(2, 5),
                   reward=torch.tensor([10.0 * t, 10.0 * t + 1]), done=torch.zeros(2), step_left=torch.full((2,), 100.0),
                   expert_action=torch.zeros(2, 5), log_probs=torch.zeros(2, 1), value=torch.zeros(2, 1))
    s = b.sample()
    assert len(b) == 6 and s["obs"].shape == (6, 1, 4, 4)
    np.testing.assert_array_equal(s["reward"].numpy(), [0, 1, 10, 11, 20, 21])        # index t*N + n
    with pytest.raises(IndexError):
        b.add_step(state=torch.zeros(2))
    b.reset()
    assert len(b) == 0


def test_ppo_bullet_trainer_runs_end_to_end_on_the_synthetic_arm():
    torch.manual_seed(0)
    np.random.seed(0)
    agent = ppoBullet(num_processes=2, total_steps=6, num_minibatches=3, update_epochs=2, target_kl=1e9, ops=oracle_ops,
                      clip_vloss=True)
    agent.device = torch.device("cpu")
    tr = ppoBulletTrainer(agent, anneal_lr=True, total_time_steps=2 * 6 * 3, num_env_steps=6, num_processes=2,
                          pretrain_episodes=2, num_eval_episodes=1)
    tr.device = torch.device("cpu")
    pi, critic = Pi(), nn.Sequential(nn.Flatten(), nn.Linear(2 * 8 * 8, 1))
    w0, c0 = pi.f[1].weight.clone(), critic[1].weight.clone()
    last = tr.run(None, {"obs_size": 8}, {}, "Synthetic-arm", pi, critic, "cnn", log=False)
    assert tr.num_updates == 3 and tr.global_step == 36
    assert last.shape == (6, 3) and np.isfinite(last).all()            # 2 epochs x 3 minibatches of the last update
    assert not torch.equal(w0, pi.f[1].weight) and not torch.equal(c0, critic[1].weight)
    assert len(tr.replay_buffer) == 0                                   # reset after every update
    np.testing.assert_allclose(agent.pi_optimizer.param_groups[0]["lr"], (1 - 2 / 3) * 1e-3)   # annealed actor lr
    assert any(t == "charts/eval_discounted_episodic_return" for (t, _, _) in tr.writer.scalars)


def test_list_of_transitions_and_dense_batch_load_identically():
    import collections
    T = collections.namedtuple("T", "state obs action reward done step_left value expert_action log_probs")
    rs = np.random.RandomState(0)
    rows = [T(int(rs.randint(2)), (rs.rand(8, 8) * 255).astype(np.float32), rs.randn(5).astype(np.float32),
              np.float32(rs.rand()), int(rs.rand() < 0.3), 5, np.float32(rs.randn()), rs.randn(5).astype(np.float32),
              np.float32(rs.randn())) for _ in range(8)]
    a = ppoBullet(num_processes=2, total_steps=4, num_minibatches=2, ops=oracle_ops)
    b = ppoBullet(num_processes=2, total_steps=4, num_minibatches=2, ops=oracle_ops)
    a._loadBatchToDevice(rows, device="cpu")
    st = lambda f: torch.as_tensor(np.stack([np.asarray(getattr(r, f)) for r in rows]))
    dense = dict(state=st("state").float(), obs=st("obs").unsqueeze(1), action=st("action"), reward=st("reward"),
                 done=st("done").float(), step_left=st("step_left").float(), value=st("value"),
                 expert_action=st("expert_action"), log_probs=st("log_probs"))
    b._loadBatchToDevice(dense, device="cpu")
    for k in ("states", "obs", "actions", "rewards", "non_final_masks", "values", "expert_actions", "log_probs"):
        np.testing.assert_allclose(a.loss_calc_dict[k].float().numpy(), b.loss_calc_dict[k].float().numpy(), err_msg=k)
    assert a.loss_calc_dict["batch_size"] == b.loss_calc_dict["batch_size"] == 4


# ---------------------------------------------------------------------------------- resume
def _ppo_params(**over):
    p = dict(gym_id="Synthetic-v0", seed=1.0, num_steps=8, gae=True, total_timesteps=8 * 4 * 4, anneal_lr=True,
             gae_lambda=0.95, num_update_epochs=2, num_envs=4, num_minibatches=2, entropy_coeff=0.0, value_coeff=0.5,
             clip_coeff=0.2, clip_vloss=True, max_grad_norm=0.5, target_kl=None, norm_adv=True, capture_video=False,
             hidden_dim=16, continuous=True, learning_rate=3e-4, exp_name="t", num_layers=2, dropout=0.0, gamma=0.99,
             track=False, log=False, save=False, device="cpu", obs_dim=5, act_dim=2)
    p.update(over)
    return p


def test_checkpoint_resume_continues_the_same_run(tmp_path):
    """Four updates straight == two updates, checkpoint, a NEW trainer resumed from it, two more: same weights, same
    Adam moments, same shuffle stream (the reference saves weights only and cannot resume, src/ppo.py:296)."""
    from aur_ppo_amd.ppo import ppo
    from tests.util import synth_rollout
    d = synth_rollout(8, 4, 5, 2, seed=3)

    def drive(agent, u0, u1):
        for u in range(u0, u1):
            agent.set_lr((1 - u / 4) * 3e-4)
            agent.buffer.rewards.copy_(torch.from_numpy(d["rewards"]) + 0.1 * u)
            ret, adv = agent.advantages(torch.from_numpy(d["next_obs"]), torch.from_numpy(d["next_done"]))
            agent.update(ret, adv)

    def fresh():
        torch.manual_seed(5)
        a = ppo(_ppo_params(), ops=oracle_ops)
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
        c.bucket.flat_param[:c.bucket.numel].add_(1.0)          # must be overwritten by the checkpoint
    assert c.load_checkpoint(path) == 2
    drive(c, 2, 4)
    torch.testing.assert_close(c.bucket.flat_param, a.bucket.flat_param, rtol=0, atol=0)
    ka, pa = a.rng.get_state()
    kc, pc = c.rng.get_state()
    np.testing.assert_array_equal(ka, kc)
    assert pa == pc
    sd = torch.load(path, weights_only=False)
    assert {"policy_state", "optimizer_state", "trainer_state", "update"} <= set(sd)


class _EpisodeEnv:
    """Fake arm runner with REAL episode semantics (what SyntheticArmEnv hides by redrawing every step): env i's
    observation carries its step-in-episode count; an episode ends after ``length[i]`` steps; stepping a finished env
    without a reset raises, as a terminal bulletarm env would misbehave."""
    device_native = False

    def __init__(self, n, lengths):
        self.n, self.lengths = n, lengths
        self.t = np.zeros(n, dtype=np.int64)
        self.finished = np.zeros(n, dtype=bool)
        self.resets = 0
        self.stepped_past_terminal = False

    def _obs(self):
        o = torch.zeros(self.n, 1, 8, 8)
        for i in range(self.n):
            o[i] = float(self.t[i])
        return torch.zeros(self.n), o

    def reset(self):
        self.t[:] = 0
        self.finished[:] = False
        return self._obs()

    def getNextAction(self):
        return torch.zeros(self.n, 5)

    def step(self, actions, auto_reset=False):
        if self.finished.any():
            self.stepped_past_terminal = True
        self.t += 1
        dones = torch.tensor([float(self.t[i] >= self.lengths[i]) for i in range(self.n)])
        rewards = dones.clone()
        for i in range(self.n):
            if dones[i] > 0:
                if auto_reset:
                    self.t[i] = 0
                    self.resets += 1
                else:
                    self.finished[i] = True
        states, obs = self._obs()
        return states, obs, rewards, dones

    def close(self):
        pass


def test_step_env_resets_finished_episodes_before_the_next_act_and_evaluates_on_upstream_cadence():
    """src/trainer/ppoBulletTrainer.py:78-85 resets the done envs after every step, :166-168 / :177-178 evaluate during the
    run.  A done env must come back as a FRESH episode (observation counter 0) in what step_env returns."""
    torch.manual_seed(0)
    np.random.seed(0)
    agent = ppoBullet(num_processes=2, total_steps=500, num_minibatches=2, update_epochs=1, target_kl=1e9, ops=oracle_ops,
                      clip_vloss=True)
    agent.device = torch.device("cpu")
    tr = ppoBulletTrainer(agent, total_time_steps=2 * 500, num_env_steps=500, num_processes=2, pretrain_episodes=0,
                          num_eval_episodes=1)
    tr.device = torch.device("cpu")
    tr.do_pretraining = False
    env = _EpisodeEnv(2, [3, 5])
    tr.initialize_env = lambda *a, **k: (setattr(tr, "envs", env), setattr(tr, "eval_envs", _EpisodeEnv(1, [2])))
    seen = []
    act0 = agent.act

    def spy_act(s, o, deterministic=False):
        if not deterministic:
            seen.append(o[:, 0, 0, 0].clone())
        return act0(s, o, deterministic=deterministic)

    agent.act = spy_act
    pi, critic = Pi(), nn.Sequential(nn.Flatten(), nn.Linear(2 * 8 * 8, 1))
    tr.run(None, {"obs_size": 8}, {}, "Synthetic-arm", pi, critic, "cnn", log=False)
    assert not env.stepped_past_terminal and env.resets == 500 // 3 + 500 // 5
    steps = torch.stack(seen)                     # (500, 2): the step-in-episode counter the policy saw
    assert steps[:, 0].max() == 2 and steps[:, 1].max() == 4          # never an observation from beyond the terminal step
    assert torch.equal(steps[:7, 0], torch.tensor([0., 1., 2., 0., 1., 2., 0.]))
    evals = [g for (t, _, g) in tr.writer.scalars if t == "charts/eval_discounted_episodic_return"]
    assert evals == [0, 1000]                     # before the run, and after the update that ends on global_step % 1000 == 0
