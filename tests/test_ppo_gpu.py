"""GPU: the trainer on the real HIP path vs traces of the real reference and vs the CPU oracle."""
import numpy as np
import pytest
import torch

from tests.util import load, synth_rollout

pytestmark = pytest.mark.gpu


def _agent(hp):
    from aur_ppo_amd.ppo import ppo
    assert torch.cuda.is_available()
    return ppo(hp)        # default ops = the HIP module


@pytest.mark.parametrize("name", ["cfg1_discrete", "cfg2_continuous", "cfg4_normal_adv_tail_clipv",
                                  "cfg3_normal_adv_tail", "cfg5_wide_128x3", "cfg6_discrete_96x1"])
def test_gpu_update_reproduces_reference_trace(name):
    # cfg3 runs clip_vloss=False, where upstream regresses the critic to its OWN old values
    # (src/ppo.py:261): that gradient is rounding noise (value_loss ~1e-15) which Adam
    # normalises, so the critic trajectory is implementation-defined (it differs between any two
    # GEMM libraries).  There only update 0's inputs, the policy-side scalars and the actor are held
    # to the tolerance; cfg4 is the same ragged-tail / discounted-return path with a conditioned loss.
    noisy_critic = name == "cfg3_normal_adv_tail"
    z = load("trace.npz")
    hp = dict(eval(str(z[f"{name}/params"])))
    init = {k[len(name) + 6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{name}/init/")}
    hp.update(gym_id="Synthetic-v0", obs_dim=init["actor.net.0.weight"].shape[1],
              act_dim=init[f"actor.net.{2 * hp['num_layers']}.weight"].shape[0], log=False, save=False)
    agent = _agent(hp)
    assert agent._mlp is not None and agent._mlp["wide"] == (name in ("cfg5_wide_128x3", "cfg6_discrete_96x1"))   # K7 / K7w, never the per-op path
    agent.policy.load_state_dict(init)
    agent.bucket.check_attached()
    agent.seed_all(1)
    U = int(z[f"{name}/num_updates"][0])
    ref_sc = z[f"{name}/scalars"]
    from aur_ppo_amd.scalars import ScalarRecorder
    w = ScalarRecorder()
    for u in range(U):
        agent.optimizer.param_groups[0]["lr"] = (1.0 - u / U) * hp["learning_rate"]
        for k in ("states", "actions", "log_probs", "rewards", "terminals", "values"):
            getattr(agent.buffer, k).copy_(torch.from_numpy(z[f"{name}/u{u}/{k}"]))
        ret, adv = agent.advantages(torch.from_numpy(z[f"{name}/u{u}/next_obs"]).cuda(),
                                    torch.from_numpy(z[f"{name}/u{u}/next_done"]).cuda())
        if not (noisy_critic and u > 0):
            np.testing.assert_allclose(adv.cpu().numpy(), z[f"{name}/u{u}/advantages"], rtol=0, atol=1e-5)
            np.testing.assert_allclose(ret.cpu().numpy(), z[f"{name}/u{u}/returns"], rtol=0, atol=1e-5)
        n = agent.update(ret, adv)
        agent._log_update(w, ret, n, (u + 1) * agent.batch_size, 0.0)
        got = [w.series(t)[-1][1] for t in ("losses/value_loss", "losses/policy_loss", "losses/entropy",
                                            "losses/old_approx_kl", "losses/approx_kl", "losses/clipfrac",
                                            "losses/explained_variance")]
        # losses within 1e-5 (north_star); clipfrac is a count ratio and may move by one sample
        if noisy_critic and u > 0:
            np.testing.assert_allclose(got[1:5], ref_sc[u][2:6], rtol=1e-2, atol=1e-4)
            continue
        np.testing.assert_allclose(got[:5], ref_sc[u][1:6], rtol=1e-4, atol=1e-5)
        assert abs(got[5] - ref_sc[u][6]) <= 1.5 / agent.minibatch_size
        np.testing.assert_allclose(got[6], ref_sc[u][7], rtol=1e-4, atol=1e-5)
    for k, v in agent.policy.state_dict().items():
        if noisy_critic:
            tol = dict(rtol=0, atol=2e-3) if k.startswith("critic") else dict(rtol=1e-3, atol=1e-5)
        else:
            tol = dict(rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(v.cpu().numpy(), z[f"{name}/final/{k}"], err_msg=k, **tol)


def test_gpu_update_vs_cpu_oracle_update_config2_shape():
    """Synthetic continuous obs 64 / act 6 (BASELINE configs[1] shape at N=64): one full update on
    the HIP path vs the oracle's reference-faithful CPU update, same init, same buffers, seed 1."""
    from oracle import ppo_oracle as O
    T, N, Dm, A = 32, 64, 64, 6
    hp = dict(gym_id="Synthetic-v0", seed=1.0, num_steps=T, gae=True, total_timesteps=T * N, anneal_lr=False,
              gae_lambda=0.95, num_update_epochs=4, num_envs=N, num_minibatches=4, entropy_coeff=0.0,
              value_coeff=0.5, clip_coeff=0.2, clip_vloss=True, max_grad_norm=0.5, target_kl=None, norm_adv=True,
              capture_video=False, hidden_dim=64, continuous=True, learning_rate=3e-4, exp_name="t", num_layers=2,
              dropout=0.0, gamma=0.99, track=False, log=False, save=False, obs_dim=Dm, act_dim=A)
    torch.manual_seed(1)
    agent = _agent(hp)
    cpu_net = O.make_actor_critic(Dm, (A,), 64, 2, True)
    cpu_net.load_state_dict({k: v.cpu() for k, v in agent.policy.state_dict().items()})
    d = synth_rollout(T, N, Dm, A)
    with torch.no_grad():   # log-probs of the stored actions under the initial policy, as in real training
        _, lp, _, _ = cpu_net.evaluate(torch.from_numpy(d["states"]).view(-1, Dm), torch.from_numpy(d["actions"]).view(-1, A))
    d["log_probs"] = (lp.view(T, N) + 0.05 * torch.randn(T, N)).numpy()
    buf = {k: torch.from_numpy(d[k]) for k in ("states", "actions", "log_probs", "rewards", "terminals", "values")}
    for k, v in buf.items():
        getattr(agent.buffer, k).copy_(v)
    agent.seed_all(1)
    ret, adv = agent.advantages(torch.from_numpy(d["next_obs"]).cuda(), torch.from_numpy(d["next_done"]).cuda())
    n = agent.update(ret, adv)
    torch.cuda.synchronize()
    opt = torch.optim.Adam(cpu_net.parameters(), lr=3e-4, eps=1e-5)
    res = O.reference_update(cpu_net, opt, buf, torch.from_numpy(d["next_obs"]), torch.from_numpy(d["next_done"]), hp,
                             np.random.RandomState(1))
    np.testing.assert_allclose(adv.cpu().numpy(), res["advantages"].numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), res["returns"].numpy(), rtol=0, atol=1e-5)
    got = agent._scalars[:n].cpu().numpy()
    assert n == 16
    cols = [0, 1, 2, 3, 4, 5, 7, 8]   # all but clipfrac
    np.testing.assert_allclose(got[:, cols], res["scalars"][:, cols], rtol=1e-4, atol=1e-5)
    assert np.abs(got[:, 6] - res["scalars"][:, 6]).max() <= 1.5 / agent.minibatch_size
    for (k, v), (_, v2) in zip(agent.policy.state_dict().items(), cpu_net.state_dict().items()):
        np.testing.assert_allclose(v.cpu().numpy(), v2.numpy(), rtol=1e-4, atol=2e-6, err_msg=k)


def test_gpu_train_end_to_end_synthetic_and_cartpole():
    hp = dict(gym_id="Synthetic-v0", seed=1.0, num_steps=32, gae=True, total_timesteps=3 * 32 * 256, anneal_lr=True,
              gae_lambda=0.95, num_update_epochs=4, num_envs=256, num_minibatches=4, entropy_coeff=0.0,
              value_coeff=0.5, clip_coeff=0.2, clip_vloss=True, max_grad_norm=0.5, target_kl=None, norm_adv=True,
              capture_video=False, hidden_dim=64, continuous=True, learning_rate=3e-4, exp_name="t", num_layers=2,
              dropout=0.0, gamma=0.99, track=False, log=False, save=False)
    a = _agent(hp)
    a.train()
    assert a.last_update["scalars"].shape == (16, 9) and np.isfinite(a.last_update["scalars"]).all()
    assert np.isfinite(a.last_update["grad_norms"]).all()
    # plumbing config (BASELINE configs[0]): CartPole-v1, 4 envs, discrete; a short run must improve
    hp2 = dict(hp, gym_id="CartPole-v1", num_envs=4, num_steps=128, total_timesteps=512 * 60, continuous=False,
               entropy_coeff=0.01, learning_rate=2.5e-4)
    b = _agent(hp2)
    r, l, x = b.train()
    assert len(r) > 20
    assert np.mean(r[-10:]) > 2.0 * np.mean(r[:10]), (np.mean(r[:10]), np.mean(r[-10:]))


def test_split_batch_linear_matches_nn_linear():
    from aur_ppo_amd.nets import _Linear
    torch.manual_seed(0)
    a, b = _Linear(64, 64).cuda(), torch.nn.Linear(64, 64).cuda()
    b.load_state_dict(a.state_dict())
    x = torch.randn(16384, 64, device="cuda")
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    w = torch.randn(16384, 64, device="cuda")
    (a(xa) * w).sum().backward()
    (b(xb) * w).sum().backward()
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-5, atol=1e-6)
    # 16384-term fp32 sums in a different order: compare at the scale of the result
    scale = float(b.weight.grad.abs().max())
    assert float((a.weight.grad - b.weight.grad).abs().max()) <= 2e-5 * scale
    assert float((a.bias.grad - b.bias.grad).abs().max()) <= 2e-5 * float(b.bias.grad.abs().max())


def test_graph_replay_matches_eager_update():
    """Same seeds, same buffers: four updates with the update captured as a hipGraph (eager, capture,
    replay, replay) end in the same weights and scalars as four eager updates."""
    T, N, Dm, A = 16, 256, 16, 3
    hp = dict(gym_id="Synthetic-v0", seed=1.0, num_steps=T, gae=True, total_timesteps=T * N * 4, anneal_lr=True,
              gae_lambda=0.95, num_update_epochs=2, num_envs=N, num_minibatches=4, entropy_coeff=0.01,
              value_coeff=0.5, clip_coeff=0.2, clip_vloss=True, max_grad_norm=0.5, target_kl=None, norm_adv=True,
              capture_video=False, hidden_dim=64, continuous=True, learning_rate=3e-4, exp_name="t", num_layers=2,
              dropout=0.0, gamma=0.99, track=False, log=False, save=False, obs_dim=Dm, act_dim=A)
    d = synth_rollout(T, N, Dm, A)
    outs = []
    for use_graph in (True, False):
        torch.manual_seed(3)
        agent = _agent(dict(hp, hip_graph=use_graph))
        for k in ("states", "actions", "log_probs", "rewards", "terminals", "values"):
            getattr(agent.buffer, k).copy_(torch.from_numpy(d[k]))
        agent.seed_all(1)
        sc = []
        for u in range(4):
            agent.set_lr((1 - u / 4) * 3e-4)
            agent.buffer.rewards.add_(0.01 * u)         # the graph must see in-place buffer changes
            ret, adv = agent.advantages(torch.from_numpy(d["next_obs"]).cuda(), torch.from_numpy(d["next_done"]).cuda())
            n = agent.update(ret, adv)
            sc.append(agent._scalars[:n].clone())
        assert (agent._graph is not None) == use_graph
        outs.append((torch.stack(sc), agent.bucket.flat_param.clone()))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-7)


def test_single_process_update_takes_the_chained_entry_point(monkeypatch):
    """Which native path the trainer takes is part of what is benchmarked: an MLP policy in one process must go
    through aurppo_mlp_ppo_minibatch_f32 (three launches per minibatch), once per minibatch, and nothing else."""
    from aur_ppo_amd import hip_ops as H
    calls = {"minibatch": 0, "step": 0, "clip_adam": 0}
    real = {k: getattr(H, k) for k in ("mlp_ppo_minibatch", "mlp_ppo_step", "clip_adam_")}

    def spy(name, key):
        def f(*a, **kw):
            calls[key] += 1
            return real[name](*a, **kw)
        return f
    monkeypatch.setattr(H, "mlp_ppo_minibatch", spy("mlp_ppo_minibatch", "minibatch"))
    monkeypatch.setattr(H, "mlp_ppo_step", spy("mlp_ppo_step", "step"))
    monkeypatch.setattr(H, "clip_adam_", spy("clip_adam_", "clip_adam"))
    hp = dict(gym_id="Synthetic-v0", seed=1.0, num_steps=16, gae=True, total_timesteps=16 * 64 * 2, anneal_lr=True,
              gae_lambda=0.95, num_update_epochs=3, num_envs=64, num_minibatches=4, entropy_coeff=0.0, value_coeff=0.5,
              clip_coeff=0.2, clip_vloss=True, max_grad_norm=0.5, target_kl=None, norm_adv=True, capture_video=False,
              hidden_dim=64, continuous=True, learning_rate=3e-4, exp_name="t", num_layers=2, dropout=0.0, gamma=0.99,
              track=False, log=False, save=False, obs_dim=16, act_dim=3, hip_graph=False)
    agent = _agent(hp)
    assert agent._mlp is not None and agent._bucket_is_policy and agent._fused_adam
    roll = synth_rollout(16, 64, 16, 3, continuous=True, seed=5)
    for k in ("states", "actions", "log_probs", "rewards", "terminals", "values"):
        getattr(agent.buffer, k).copy_(torch.from_numpy(roll[k]))
    ret, adv = agent.advantages(torch.from_numpy(roll["next_obs"]).cuda(), torch.from_numpy(roll["next_done"]).cuda())
    n = agent.update(ret, adv)
    torch.cuda.synchronize()
    assert n == 12 and calls == {"minibatch": 12, "step": 0, "clip_adam": 0}
    assert float(agent._adam_t) == 12.0 and torch.isfinite(agent.bucket.flat_param).all()


def test_captured_rollout_matches_eager_rollout():
    """train() on the device-resident synthetic env: rollouts replayed from a hipGraph (capturable env + K8) leave the
    same policy as rollouts run step by step -- same generators, same draws, same buffer contents."""
    def run(graph):
        hp = dict(gym_id="Synthetic-v0", seed=1.0, num_steps=8, gae=True, total_timesteps=8 * 64 * 5, anneal_lr=True,
                  gae_lambda=0.95, num_update_epochs=2, num_envs=64, num_minibatches=2, entropy_coeff=0.0, value_coeff=0.5,
                  clip_coeff=0.2, clip_vloss=True, max_grad_norm=0.5, target_kl=None, norm_adv=True, capture_video=False,
                  hidden_dim=64, continuous=True, learning_rate=3e-4, exp_name="t", num_layers=2, dropout=0.0, gamma=0.99,
                  track=False, log=False, save=False, obs_dim=16, act_dim=3, hip_graph=graph)
        torch.manual_seed(7)
        agent = _agent(hp)
        agent.train()
        torch.cuda.synchronize()
        return agent, agent.bucket.flat_param.clone(), agent.buffer.states.clone(), agent.buffer.actions.clone()

    a_g, p_g, s_g, act_g = run(True)
    a_e, p_e, s_e, act_e = run(False)
    assert a_g._ro_state == 2 and a_g._ro_graph is not None and a_e._ro_graph is None
    assert torch.equal(s_g, s_e)                       # the env's generator advanced identically
    torch.testing.assert_close(act_g, act_e, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(p_g, p_e, rtol=1e-4, atol=1e-6)
