"""CPU: host logic of the trainer (API surface, minibatch slicing, LR anneal, logging, early
stop) with the oracle standing in for the kernels, checked against traces of the real reference."""
import numpy as np
import pytest
import torch

from aur_ppo_amd.ppo import ppo, torch_buffer
from tests import oracle_ops
from tests.util import load


def _params(**over):
    p = dict(gym_id="Synthetic-v0", seed=1.0, num_steps=16, gae=True, total_timesteps=256, anneal_lr=True,
             gae_lambda=0.95, num_update_epochs=4, num_envs=8, num_minibatches=4, entropy_coeff=0.01,
             value_coeff=0.5, clip_coeff=0.2, clip_vloss=True, max_grad_norm=0.5, target_kl=None, norm_adv=True,
             capture_video=False, hidden_dim=64, continuous=False, learning_rate=2.5e-4, exp_name="t",
             num_layers=2, dropout=0.0, gamma=0.99, track=False, log=False, save=False, device="cpu")
    p.update(over)
    return p


def _agent_from_trace(z, name):
    hp = dict(eval(str(z[f"{name}/params"])))
    init = {k[len(name) + 6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{name}/init/")}
    D = init["actor.net.0.weight"].shape[1]
    A = init[f"actor.net.{2 * hp['num_layers']}.weight"].shape[0]
    hp.update(gym_id="Synthetic-v0", obs_dim=D, act_dim=A, log=False, save=False, device="cpu")
    agent = ppo(hp, ops=oracle_ops)
    agent.policy.load_state_dict(init)
    agent.bucket.check_attached()
    return agent, hp


def test_api_surface_matches_reference():
    a = ppo(_params(), ops=oracle_ops)
    for attr in ("all_steps", "batch_size", "minibatch_size", "num_updates", "run_name", "envs", "state_dim",
                 "action_dim", "policy", "buffer", "optimizer", "total_returns", "total_episode_lengths", "x_indices"):
        assert hasattr(a, attr), attr
    for m in ("make_env", "rewards_to_go", "run_gae", "normal_advantage", "advantages", "train", "plot", "moving_average",
              "plot_episodic_returns"):
        assert callable(getattr(a, m)), m
    assert (a.batch_size, a.minibatch_size, a.num_updates) == (128, 32, 2)
    assert isinstance(a.buffer, torch_buffer) and a.buffer.states.shape == (16, 8, 4)
    assert a.optimizer.defaults["eps"] == 1e-5
    sd = a.policy.state_dict()
    assert list(sd) == ["actor.net.0.weight", "actor.net.0.bias", "actor.net.2.weight", "actor.net.2.bias",
                        "actor.net.4.weight", "actor.net.4.bias", "critic.net.0.weight", "critic.net.0.bias",
                        "critic.net.2.weight", "critic.net.2.bias", "critic.net.4.weight", "critic.net.4.bias"]
    b = a.buffer.flatten(torch.zeros(16, 8), torch.ones(16, 8))
    assert [tuple(t.shape) for t in b] == [(128, 4), (128,), (128,), (128,), (128,), (128,)]
    assert b[0].data_ptr() == a.buffer.states.data_ptr()      # views, not copies


def test_make_env_returns_the_reference_thunk(monkeypatch):
    """src/ppo.py:85-99: make_env returns a thunk; calling it wraps gym.make(gym_id) with RecordEpisodeStatistics, RecordVideo for
    env 0 when asked, and -- continuous control only -- ClipAction, NormalizeObservation, TransformObservation, NormalizeReward,
    TransformReward, in that order.  gym is not in this image: a recording stand-in module checks the order."""
    import sys
    import types
    calls = []

    def wrapper(name):
        def w(env, *a):
            calls.append(name)
            return ("wrapped", name, env)
        return w
    fake = types.ModuleType("gym")
    fake.make = lambda gid: (calls.append(f"make:{gid}"), ("env", gid))[1]
    fake.wrappers = types.SimpleNamespace(**{n: wrapper(n) for n in ("RecordEpisodeStatistics", "RecordVideo", "ClipAction",
                                                                       "NormalizeObservation", "TransformObservation",
                                                                       "NormalizeReward", "TransformReward")})
    a = ppo(_params(), ops=oracle_ops)
    thunk = a.make_env("CartPole-v1", 0, True)
    assert callable(thunk) and calls == []            # nothing is built until the thunk runs
    monkeypatch.setitem(sys.modules, "gym", fake)
    thunk()
    assert calls == ["make:CartPole-v1", "RecordEpisodeStatistics", "RecordVideo"]
    calls.clear()
    a.make_env("CartPole-v1", 1, True)()
    assert calls == ["make:CartPole-v1", "RecordEpisodeStatistics"]
    calls.clear()
    a.continuous = True
    a.make_env("Hopper-v4", 1, False)()
    assert calls == ["make:Hopper-v4", "RecordEpisodeStatistics", "ClipAction", "NormalizeObservation", "TransformObservation",
                     "NormalizeReward", "TransformReward"]
    monkeypatch.delitem(sys.modules, "gym")
    monkeypatch.setitem(sys.modules, "gymnasium", None)
    with pytest.raises(ImportError):
        a.make_env("CartPole-v1", 0, False)()


@pytest.mark.parametrize("name", ["cfg1_discrete", "cfg2_continuous", "cfg3_normal_adv_tail", "cfg4_normal_adv_tail_clipv",
                                  "cfg5_wide_128x3", "cfg6_discrete_96x1"])
def test_trainer_update_reproduces_reference_trace(name):
    torch.set_num_threads(1)
    z = load("trace.npz")
    agent, hp = _agent_from_trace(z, name)
    agent.seed_all(1)
    U = int(z[f"{name}/num_updates"][0])
    assert agent.num_updates == U
    ref_sc = z[f"{name}/scalars"]
    from aur_ppo_amd.scalars import ScalarRecorder
    w = ScalarRecorder()
    for u in range(U):
        agent.optimizer.param_groups[0]["lr"] = (1.0 - u / U) * hp["learning_rate"]
        for k in ("states", "actions", "log_probs", "rewards", "terminals", "values"):
            getattr(agent.buffer, k).copy_(torch.from_numpy(z[f"{name}/u{u}/{k}"]))
        ret, adv = agent.advantages(torch.from_numpy(z[f"{name}/u{u}/next_obs"]),
                                    torch.from_numpy(z[f"{name}/u{u}/next_done"]))
        # (bootstrap value comes from a CPU GEMM on flat-bucket views: last-bit differences allowed)
        np.testing.assert_allclose(adv.numpy(), z[f"{name}/u{u}/advantages"], rtol=0, atol=2e-6)
        n = agent.update(ret, adv)
        assert n == hp["num_update_epochs"] * int(np.ceil(agent.batch_size / agent.minibatch_size))
        agent._log_update(w, ret, n, (u + 1) * agent.batch_size, 0.0)
        got = [w.series(t)[-1][1] for t in ("losses/value_loss", "losses/policy_loss", "losses/entropy",
                                            "losses/old_approx_kl", "losses/approx_kl", "losses/clipfrac",
                                            "losses/explained_variance")]
        np.testing.assert_allclose(got, ref_sc[u][1:8], rtol=2e-5, atol=2e-7)
    for k, v in agent.policy.state_dict().items():
        np.testing.assert_allclose(v.numpy(), z[f"{name}/final/{k}"], rtol=1e-5, atol=1e-7, err_msg=k)


def test_train_runs_end_to_end_and_logs_reference_tags():
    a = ppo(_params(continuous=True, obs_dim=5, act_dim=3, entropy_coeff=0.0), ops=oracle_ops)
    r, l, x = a.train()
    assert (r, l, x) == ([], [], [])
    tags = {t for (t, _, _) in a.writer.scalars}
    assert {"charts/learning_rate", "losses/value_loss", "losses/policy_loss", "losses/entropy",
            "losses/old_approx_kl", "losses/approx_kl", "losses/clipfrac", "losses/explained_variance",
            "charts/SPS"} <= tags
    lrs = [v for (_, v) in a.writer.series("charts/learning_rate")]
    np.testing.assert_allclose(lrs, [2.5e-4, 1.25e-4])            # src/ppo.py:195-198
    assert all(np.isfinite(v) for (_, v, _) in a.writer.scalars)


def test_target_kl_early_stop_rewinds_shuffle_stream():
    # with target_kl=0 every update stops after its first epoch; upstream would then have drawn
    # exactly one shuffle per update from the global stream
    a = ppo(_params(target_kl=0.0, total_timesteps=384), ops=oracle_ops)
    a.train()
    assert a.last_update["scalars"].shape[0] == 4                 # one epoch x 4 minibatches
    rs = np.random.RandomState(1)
    for _ in range(3):
        b = np.arange(128)
        rs.shuffle(b)
    key, pos = a.rng.get_state()
    st = rs.get_state()
    np.testing.assert_array_equal(key, st[1])
    assert pos == st[2]


def test_cartpole_builtin_env_learns_a_little():
    torch.manual_seed(1)
    a = ppo(_params(gym_id="CartPole-v1", num_envs=4, num_steps=128, total_timesteps=4 * 128 * 12), ops=oracle_ops)
    r, l, x = a.train()
    assert len(r) > 5 and all(1 <= v <= 500 for v in r) and l[0] == r[0]
    assert np.mean(r[-5:]) > np.mean(r[:5]) * 0.8      # not diverging; real learning is checked on the GPU
