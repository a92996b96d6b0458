"""GPU: the parity chain closed at the benchmarked sizes, against the oracle and the reference's own fixtures.

  * one full update of ``aur_ppo_amd.ppo`` (default ops: K8 bootstrap, K1, K2, K7 chain, K6b) at BASELINE configs 4 and 2
    (N=4096 / N=1024, T=128, D=64, A=6, E=4, 4 minibatches, seed 1, SURVEY 8d generator = bench.py's) both eagerly
    and as a captured hipGraph, vs ``oracle.reference_update`` (src/ppo.py:125-142,159-166,213-269) on the same tensors;
  * K7 / K8 against tests/golden/evaluate.npz (outputs and autograd gradients of the REAL reference's
    ``actor_critic.evaluate``, src/models/actor_critic.py:34-51);
  * the ``target_kl`` early stop (src/ppo.py:271-273) through the device RNG rewind, on the GPU trainer.
"""
import numpy as np
import pytest
import torch

from tests.util import load

pytestmark = pytest.mark.gpu

_oracle_cache = {}


def _hp(N, T=128, Dm=64, A=6, **kw):
    hp = dict(gym_id="Synthetic-v0", seed=1.0, num_steps=T, gae=True, total_timesteps=T * N, anneal_lr=False,
              gae_lambda=0.95, num_update_epochs=4, num_envs=N, num_minibatches=4, entropy_coeff=0.0,
              value_coeff=0.5, clip_coeff=0.2, clip_vloss=True, max_grad_norm=0.5, target_kl=None, norm_adv=True,
              capture_video=False, hidden_dim=64, continuous=True, learning_rate=3e-4, exp_name="t", num_layers=2,
              dropout=0.0, gamma=0.99, track=False, log=False, save=False, obs_dim=Dm, act_dim=A)
    hp.update(kw)
    return hp


def _oracle_update(N, T, Dm, A, hp, init_sd, data):
    """oracle.reference_update on the SURVEY 8d tensors, computed once per size and net shape (a few seconds of CPU)."""
    key = (N, T, Dm, A, hp["hidden_dim"], hp["num_layers"])
    if key not in _oracle_cache:
        from oracle import ppo_oracle as O
        net = O.make_actor_critic(Dm, (A,), hp["hidden_dim"], hp["num_layers"], True)
        net.load_state_dict(init_sd)
        opt = torch.optim.Adam(net.parameters(), lr=hp["learning_rate"], eps=1e-5)
        buf = {k: data[k] for k in ("states", "actions", "log_probs", "rewards", "terminals", "values")}
        res = O.reference_update(net, opt, buf, data["next_obs"], data["next_done"], hp, np.random.RandomState(1))
        _oracle_cache[key] = (res, {k: v.detach().clone() for k, v in net.state_dict().items()})
    return _oracle_cache[key]


@pytest.mark.parametrize("launch", ["eager", "hipGraph"])
@pytest.mark.parametrize("N", [1024, 4096], ids=["config2_N1024", "config4_N4096"])
def test_full_update_at_bench_size_matches_oracle(N, launch):
    import bench
    from aur_ppo_amd.ppo import ppo
    T, Dm, A = 128, 64, 6
    hp = _hp(N, T, Dm, A, hip_graph=(launch == "hipGraph"))
    torch.manual_seed(1)
    agent = ppo(hp)
    assert agent._mlp is not None and agent._bucket_is_policy and agent._fused_adam      # the benchmarked path
    data = bench.synth_buffers(T, N, Dm, A, 1234)
    init_sd = {k: v.detach().cpu().clone() for k, v in agent.policy.state_dict().items()}
    for k in ("states", "actions", "values", "rewards", "terminals"):
        getattr(agent.buffer, k).copy_(data[k])
    with torch.no_grad():     # old log-probs = the policy's own at init weights
        _, lp, _, _ = agent.policy.evaluate(agent.buffer.states.view(-1, Dm), agent.buffer.actions.view(-1, A))
        agent.buffer.log_probs.copy_(lp.view(T, N))
    data["log_probs"] = agent.buffer.log_probs.cpu()
    agent.seed_all(1)
    if launch == "hipGraph":
        agent._graph_state = 1       # capture on this very update (the flat Adam state exists already) and replay it
    ret, adv = agent.advantages(data["next_obs"].cuda(), data["next_done"].cuda())
    n = agent.update(ret, adv)
    torch.cuda.synchronize()
    assert (agent._graph is not None) == (launch == "hipGraph")
    assert n == 16
    res, final_sd = _oracle_update(N, T, Dm, A, hp, init_sd, data)
    # bit-exact permutations (integer work)
    perms = agent._last_perms.cpu().numpy()
    for e in range(4):
        assert np.array_equal(perms[e], res["perms"][e]), f"epoch {e} permutation"
    # advantages / returns within 1e-5 (north_star)
    np.testing.assert_allclose(adv.cpu().numpy(), res["advantages"].numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), res["returns"].numpy(), rtol=0, atol=1e-5)
    # all 16 rows of the 9 loss scalars (tolerances of test_ppo_gpu.py)
    got = agent._scalars[:n].cpu().numpy()
    cols = [0, 1, 2, 3, 4, 5, 7, 8]   # all but clipfrac
    np.testing.assert_allclose(got[:, cols], res["scalars"][:, cols], rtol=1e-4, atol=1e-5)
    assert np.abs(got[:, 6] - res["scalars"][:, 6]).max() <= 1.5 / agent.minibatch_size
    # final weights after the 16 clip + Adam steps
    for k, v in agent.policy.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), final_sd[k].numpy(), rtol=1e-4, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("tiles", ["static", "counter"])
@pytest.mark.parametrize("layers,hidden", [(3, 64), (3, 128)], ids=["3x64_both_nets_per_wg", "3x128_one_net_per_wg"])
def test_k7w_full_update_at_bench_size_matches_oracle(layers, hidden, tiles, monkeypatch):
    """The other `-d` / `-nl` net shapes (src/run_ppo.py:33,37) at the BENCHMARKED size: one whole captured update of K7w at
    N 4096 / T 128 (16 x {prepare, k_mlpw_step, reduce, clip + Adam}) against oracle.reference_update on the same tensors.
    tiles = static: AURPPO_STATIC_TILES deals the row tiles by stride, every sum has a fixed order and the final weights are
    held to the 2e-6 the default net's are.  tiles = counter (the product mode): a gradient element's last bits depend on
    the launch, Adam's lr / eps = 30 amplifies that in weights whose gradient is far below eps, and the bound is 2e-5
    (bench.py::check_parity has the derivation); every other quantity keeps its tolerance in both modes."""
    import bench
    from aur_ppo_amd.ppo import ppo
    if tiles == "static":
        monkeypatch.setenv("AURPPO_STATIC_TILES", "1")
    else:
        monkeypatch.delenv("AURPPO_STATIC_TILES", raising=False)
    N, T, Dm, A = 4096, 128, 64, 6
    hp = _hp(N, T, Dm, A, hip_graph=True, hidden_dim=hidden, num_layers=layers)
    torch.manual_seed(1)
    agent = ppo(hp)
    assert agent._mlp is not None and agent._mlp.get("wide") and agent._bucket_is_policy and agent._fused_adam
    data = bench.synth_buffers(T, N, Dm, A, 1234)
    init_sd = {k: v.detach().cpu().clone() for k, v in agent.policy.state_dict().items()}
    for k in ("states", "actions", "values", "rewards", "terminals"):
        getattr(agent.buffer, k).copy_(data[k])
    with torch.no_grad():
        _, lp, _, _ = agent.policy.evaluate(agent.buffer.states.view(-1, Dm), agent.buffer.actions.view(-1, A))
        agent.buffer.log_probs.copy_(lp.view(T, N))
    data["log_probs"] = agent.buffer.log_probs.cpu()
    agent.seed_all(1)
    agent._graph_state = 1
    ret, adv = agent.advantages(data["next_obs"].cuda(), data["next_done"].cuda())
    n = agent.update(ret, adv)
    torch.cuda.synchronize()
    assert agent._graph is not None and n == 16
    res, final_sd = _oracle_update(N, T, Dm, A, hp, init_sd, data)
    perms = agent._last_perms.cpu().numpy()
    for e in range(4):
        assert np.array_equal(perms[e], res["perms"][e]), f"epoch {e} permutation"
    np.testing.assert_allclose(adv.cpu().numpy(), res["advantages"].numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), res["returns"].numpy(), rtol=0, atol=1e-5)
    got = agent._scalars[:n].cpu().numpy()
    cols = [0, 1, 2, 3, 4, 5, 7, 8]
    np.testing.assert_allclose(got[:, cols], res["scalars"][:, cols], rtol=1e-4, atol=1e-5)
    assert np.abs(got[:, 6] - res["scalars"][:, 6]).max() <= 1.5 / agent.minibatch_size
    w_atol = 2e-6 if tiles == "static" else 2e-5
    for k, v in agent.policy.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), final_sd[k].numpy(), rtol=1e-4, atol=w_atol, err_msg=f"{k} ({tiles} tiles)")


# ---------------------------------------------------------------------------------- evaluate.npz (real reference outputs)
def _fixture_policy(name):
    from aur_ppo_amd import hip_ops as H
    from aur_ppo_amd.actor_critic import actor_critic
    from aur_ppo_amd.flat import FlatBucket
    z = load("evaluate.npz")
    D, A, cont, layers, hid = (int(x) for x in z[f"{name}/meta"])
    pol = actor_critic(D, (A,) if cont else A, hid, layers, 0.0, bool(cont)).cuda()
    pol.load_state_dict({k: torch.from_numpy(z[f"{name}/sd/{k}"]) for k in pol.state_dict()})
    bucket = FlatBucket(pol.parameters())
    lay = H.mlp_layout(pol, bucket)
    assert lay is not None
    return H, z, pol, bucket, lay, D, A, bool(cont)


@pytest.mark.parametrize("name", ["cont_D64_A6", "disc_D4_A2", "cont_D5_A3_L3", "cont_D64_A6_H128_L3", "disc_D8_A4_H128_L2", "cont_D128_A6_H96_L1"])
def test_k8_value_only_matches_reference_value_fn(name):
    H, z, pol, bucket, lay, D, A, cont = _fixture_policy(name)
    obs = torch.from_numpy(z[f"{name}/obs"]).cuda()
    _, _, v = H.mlp_act(obs, None, bucket.flat_param, lay)
    np.testing.assert_allclose(v.cpu().numpy(), z[f"{name}/value_fn"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("name", ["cont_D64_A6", "disc_D4_A2", "cont_D5_A3_L3", "cont_D64_A6_H128_L3", "disc_D8_A4_H128_L2", "cont_D128_A6_H96_L1"])
def test_k8_sampled_logp_and_value_match_reference_formulas(name):
    """K8 with noise: the action it samples, evaluated by the oracle's restatement of ``evaluate`` with the fixture's
    weights (held to evaluate.npz by tests/test_oracle_golden.py), must give K8's log-prob and value."""
    from oracle import ppo_oracle as O
    H, z, pol, bucket, lay, D, A, cont = _fixture_policy(name)
    obs = torch.from_numpy(z[f"{name}/obs"]).cuda()
    g = torch.Generator(device="cpu").manual_seed(5)
    noise = (torch.randn(37, A, generator=g) if cont else torch.rand(37, generator=g)).cuda()
    act, logp, v = H.mlp_act(obs, noise, bucket.flat_param, lay)
    _, _, _, layers, hid = (int(x) for x in z[f"{name}/meta"])
    net = O.make_actor_critic(D, (A,) if cont else A, hid, layers, cont)
    net.load_state_dict({k: torch.from_numpy(z[f"{name}/sd/{k}"]) for k in net.state_dict()})
    with torch.no_grad():
        a_cpu = act.cpu() if cont else act.cpu().long()
        _, lp_o, _, v_o = net.evaluate(obs.cpu(), a_cpu)
        if cont:     # the sample itself: mean + std * noise (torch Normal.sample)
            mean = net.actor(obs.cpu())
            np.testing.assert_allclose(act.cpu().numpy(), (mean + torch.exp(net.actor_logstd) * noise.cpu()).numpy(),
                                       rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(logp.cpu().numpy(), lp_o.numpy(), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(v.cpu().numpy(), v_o.view(-1).numpy(), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("name", ["cont_D64_A6", "disc_D4_A2", "cont_D5_A3_L3", "cont_D64_A6_H128_L3", "disc_D8_A4_H128_L2", "cont_D128_A6_H96_L1"])
def test_k7_forward_matches_reference_evaluate_per_sample(name):
    """K7 has no per-sample outputs; a minibatch of ONE sample exposes them: with old_logp = 0 the
    ``old_approx_kl`` scalar is -logp, ``entropy`` is the sample's entropy, and with the un-clipped value loss against
    a return of 0 the critic head's bias gradient is vf_coef * v."""
    H, z, pol, bucket, lay, D, A, cont = _fixture_policy(name)
    obs = torch.from_numpy(z[f"{name}/obs"]).cuda()
    act = torch.from_numpy(z[f"{name}/act"]).float().cuda().contiguous()
    B = obs.shape[0]
    rec = torch.zeros(B, 4, device="cuda")            # {old_logp, adv, ret, old_v} = 0
    rec[:, 1] = 1.0
    b3c = lay["offsets"][-2]                          # critic head bias (last entry before actor_logstd in both layouts)
    logp, ent, val = [], [], []
    g = torch.empty_like(bucket.flat_grad)
    for i in range(B):
        idx = torch.tensor([i], device="cuda", dtype=torch.int32)
        sc = H.mlp_ppo_step(obs, act, rec, idx, bucket.flat_param, lay, g, 0.2, 0.01, 1.0, False, H.VLOSS_RETURNS)
        sc = sc.cpu().numpy()
        logp.append(-sc[H.S_OLD_KL])
        ent.append(sc[H.S_ENT])
        val.append(float(g[b3c]))
    np.testing.assert_allclose(np.array(logp), z[f"{name}/logp"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(np.array(ent), z[f"{name}/ent"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(np.array(val), z[f"{name}/val"].reshape(-1), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("name", ["cont_D64_A6", "disc_D4_A2", "cont_D5_A3_L3", "cont_D64_A6_H128_L3", "disc_D8_A4_H128_L2", "cont_D128_A6_H96_L1"])
def test_k7_backward_matches_reference_autograd_gradients(name):
    """The fixture's gradients are the reference's own autograd of  sum(w*logp) + 0.3*sum(ent) + sum(val^2)  through
    ``actor_critic.evaluate``.  K7's loss becomes exactly that with old_logp = logp (ratio 1, inside the clip),
    A_i = -M*w_i without normalisation, ent_coef = -0.3*M, vf_coef = 2*M against returns of 0."""
    H, z, pol, bucket, lay, D, A, cont = _fixture_policy(name)
    obs = torch.from_numpy(z[f"{name}/obs"]).cuda()
    act = torch.from_numpy(z[f"{name}/act"]).float().cuda().contiguous()
    M = obs.shape[0]
    w = torch.linspace(0.5, 1.5, M)
    rec = torch.zeros(M, 4)
    rec[:, 0] = torch.from_numpy(z[f"{name}/logp"])
    rec[:, 1] = -M * w
    rec = rec.cuda()
    idx = torch.arange(M, device="cuda", dtype=torch.int32)
    g = torch.full_like(bucket.flat_grad, float("nan"))
    H.mlp_ppo_step(obs, act, rec, idx, bucket.flat_param, lay, g, 0.2, -0.3 * M, 2.0 * M, False, H.VLOSS_RETURNS)
    off = 0
    names = [k for k, _ in pol.named_parameters()]
    assert [id(p) for p in bucket.params] == [id(p) for _, p in pol.named_parameters()]
    for k, p in zip(names, bucket.params):
        ref = z[f"{name}/grad/{k}"].reshape(-1)
        got = g[off:off + p.numel()].cpu().numpy()
        s = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 2e-5 * s + 1e-6, (k, np.abs(got - ref).max(), s)
        off += p.numel()


# ---------------------------------------------------------------------------------- target_kl on the GPU trainer
def test_target_kl_early_stop_on_gpu_rewinds_device_rng():
    """src/ppo.py:271-273: with target_kl = 0 every update stops after its first epoch, so upstream draws ONE shuffle
    per update from numpy's global stream.  The device stream pre-draws all E: it must be rewound, the generator must
    sit where numpy's sits, and the next update must be stepping through numpy's next permutation."""
    from aur_ppo_amd.ppo import ppo
    T, N = 16, 64
    hp = _hp(N, T, 16, 3, total_timesteps=3 * T * N, target_kl=0.0, anneal_lr=True)
    torch.manual_seed(1)
    a = ppo(hp)
    a.train()
    torch.cuda.synchronize()
    assert a._graph is None                                    # target_kl keeps the update eager
    assert a.last_update["scalars"].shape[0] == 4              # one epoch x 4 minibatches
    rs = np.random.RandomState(1)
    last = None
    for _ in range(3):
        b = np.arange(T * N)
        rs.shuffle(b)
        last = b.copy()
    assert np.array_equal(a._last_perms[0].cpu().numpy(), last)     # update 3 ran on numpy's third shuffle
    key, pos = a.rng.get_state()
    st = rs.get_state()
    np.testing.assert_array_equal(key, st[1])
    assert pos == st[2]
    # a run without the early stop from the same seed must differ from epoch 2 on (the stop really happened)
    torch.manual_seed(1)
    b_ = ppo(dict(hp, target_kl=None))
    b_.train()
    assert b_.last_update["scalars"].shape[0] == 16
    # ... and a generous threshold never stops: same permutations as the unconstrained run
    torch.manual_seed(1)
    c_ = ppo(dict(hp, target_kl=1e9))
    c_.train()
    assert c_.last_update["scalars"].shape[0] == 16
    assert torch.equal(c_._last_perms, b_._last_perms)
    np.testing.assert_allclose(c_.last_update["scalars"], b_.last_update["scalars"], rtol=1e-5, atol=1e-6)
