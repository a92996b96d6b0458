"""GPU: K7w / K8w (csrc/mlp_wide.hip) -- the fused minibatch step and rollout step for the MLP shapes the reference's CLI
can ask for besides the default 64-64 (src/run_ppo.py:33,37 -d / -nl; src/nets/nets.py:19-53) -- against the per-op autograd
path (K3 gather -> torch evaluate -> K5 loss -> autograd), the same treatment K7 gets in test_mlp_fused.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(T, N, D, A, hidden, layers, seed=0, cont=True):
    from aur_ppo_amd import hip_ops as H
    from aur_ppo_amd.actor_critic import actor_critic
    from aur_ppo_amd.flat import FlatBucket
    torch.manual_seed(seed)
    pol = actor_critic(D, (A,) if cont else A, hidden, layers, 0.0, cont).cuda()
    with torch.no_grad():
        if cont:
            pol.actor_logstd.copy_(0.3 * torch.randn(1, A))
        for p in pol.parameters():          # make every layer matter (head init is 0.01-scaled)
            p.add_((0.05 if cont else 0.3) * torch.randn_like(p))
    bucket = FlatBucket(pol.parameters())
    B = T * N
    g = torch.Generator(device="cuda").manual_seed(seed)
    obs = torch.randn(B, D, device="cuda", generator=g)
    act = (torch.randn(B, A, device="cuda", generator=g) if cont
           else torch.randint(0, A, (B,), device="cuda", generator=g).float())
    with torch.no_grad():
        _, lp, _, v = pol.evaluate(obs, act)
    rec = torch.stack([lp + 0.2 * torch.randn(B, device="cuda", generator=g), 2 * torch.randn(B, device="cuda", generator=g),
                       v.view(-1) + torch.randn(B, device="cuda", generator=g),
                       v.view(-1) + 0.1 * torch.randn(B, device="cuda", generator=g)], 1).contiguous()
    return H, pol, bucket, obs, act, rec


SHAPES = [  # hidden, layers, D, A, T, N, M
    (128, 2, 64, 6, 8, 64, 256), (128, 3, 128, 6, 8, 64, 500), (64, 3, 5, 3, 8, 32, 200), (64, 1, 64, 6, 8, 32, 100),
    (32, 2, 8, 2, 8, 32, 77), (100, 2, 17, 4, 8, 64, 333), (96, 3, 33, 16, 8, 32, 31), (128, 1, 100, 1, 8, 32, 256),
    (64, 2, 128, 6, 8, 64, 512), (7, 1, 3, 1, 4, 32, 100), (128, 3, 64, 6, 64, 1024, 32768), (128, 2, 64, 6, 128, 1024, 131072),
]


@pytest.mark.parametrize("hidden,layers,D,A,T,N,M", SHAPES)
@pytest.mark.parametrize("norm_adv,vmode,packed", [(True, 1, True), (False, 2, False), (True, 0, False)])
def test_wide_step_matches_autograd_path(hidden, layers, D, A, T, N, M, norm_adv, vmode, packed):
    H, pol, bucket, obs, act, rec = _setup(T, N, D, A, hidden, layers)
    idx = torch.randperm(T * N, device="cuda")[:M].int()
    lay = H.mlp_layout(pol, bucket)
    assert lay is not None and lay["wide"] and (lay["hidden"], lay["num_layers"], lay["D"], lay["A"]) == (hidden, layers, D, A)
    mb = H.gather(idx, [obs, act, rec])
    _, nlp, ent, nv = pol.evaluate(mb[0], mb[1])
    sc_ref = torch.empty(9, device="cuda")
    loss = H.ppo_loss_packed(nlp, nv, ent, mb[2], 0.2, 0.01, 0.5, norm_adv, vmode, sc_ref)
    bucket.zero_grad()
    loss.backward()
    g_ref = bucket.flat_grad[:lay["n_params"]].clone()
    g_out = torch.full_like(bucket.flat_grad, float("nan"))
    if packed and A <= 12:
        sc = H.mlp_ppo_step(obs, None, H.pack_records(rec, act), idx, bucket.flat_param, lay, g_out, 0.2, 0.01, 0.5, norm_adv, vmode)
    else:
        sc = H.mlp_ppo_step(obs, act, rec, idx, bucket.flat_param, lay, g_out, 0.2, 0.01, 0.5, norm_adv, vmode)
    torch.cuda.synchronize()
    np.testing.assert_allclose(sc.cpu().numpy(), sc_ref.cpu().numpy(), rtol=2e-5, atol=2e-6)
    g = g_out[:lay["n_params"]]
    assert torch.isfinite(g).all()
    scale = float(g_ref.abs().max())
    assert float((g - g_ref).abs().max()) <= 2e-5 * scale + 1e-8, (float((g - g_ref).abs().max()), scale)
    off = 0
    for p_, (nm, _) in zip(bucket.params, pol.named_parameters()):
        k = p_.numel()
        a, b = g[off:off + k], g_ref[off:off + k]
        s = float(b.abs().max())
        # floor: a bias gradient is a sum of M terms that may cancel to far below the terms' own rounding error
        assert float((a - b).abs().max()) <= 5e-5 * s + 2e-6 * scale + 1e-9, (nm, float((a - b).abs().max()), s)
        off += k


@pytest.mark.parametrize("hidden,layers,D,A,T,N,M", [(128, 2, 4, 2, 8, 64, 200), (64, 3, 64, 16, 16, 64, 1024), (128, 3, 6, 11, 8, 64, 333),
                                                    (48, 1, 5, 3, 8, 64, 200)])
@pytest.mark.parametrize("norm_adv,vmode,ec", [(True, 1, 0.01), (False, 2, 0.05)])
def test_wide_step_categorical_head_matches_autograd_path(hidden, layers, D, A, T, N, M, norm_adv, vmode, ec):
    H, pol, bucket, obs, act, rec = _setup(T, N, D, A, hidden, layers, seed=1, cont=False)
    idx = torch.randperm(T * N, device="cuda")[:M].int()
    lay = H.mlp_layout(pol, bucket)
    assert lay is not None and lay["wide"] and lay["continuous"] is False
    mb = H.gather(idx, [obs, act, rec])
    _, nlp, ent, nv = pol.evaluate(mb[0], mb[1])
    sc_ref = torch.empty(9, device="cuda")
    loss = H.ppo_loss_packed(nlp, nv, ent, mb[2], 0.2, ec, 0.5, norm_adv, vmode, sc_ref)
    bucket.zero_grad()
    loss.backward()
    g_ref = bucket.flat_grad[:lay["n_params"]].clone()
    g_out = torch.full_like(bucket.flat_grad, float("nan"))
    sc = H.mlp_ppo_step(obs, act, rec, idx, bucket.flat_param, lay, g_out, 0.2, ec, 0.5, norm_adv, vmode)
    np.testing.assert_allclose(sc.cpu().numpy(), sc_ref.cpu().numpy(), rtol=2e-5, atol=2e-6)
    g = g_out[:lay["n_params"]]
    assert torch.isfinite(g).all()
    gscale = float(g_ref.abs().max())
    assert float((g - g_ref).abs().max()) <= 2e-5 * gscale + 1e-8
    off = 0
    for p_ in bucket.params:
        k = p_.numel()
        a, b = g[off:off + k], g_ref[off:off + k]
        s_ = float(b.abs().max())
        assert float((a - b).abs().max()) <= 5e-5 * s_ + 2e-6 * gscale + 1e-9, (off, float((a - b).abs().max()), s_)
        off += k


@pytest.mark.parametrize("hidden,layers,N,D,A,cont", [(128, 2, 4096, 64, 6, True), (128, 3, 77, 128, 2, False), (64, 3, 256, 16, 16, True),
                                                      (32, 1, 33, 8, 5, False), (100, 2, 1, 64, 1, True), (64, 1, 100, 11, 3, True)])
def test_wide_act_kernel_matches_torch_formulas(hidden, layers, N, D, A, cont):
    H, pol, bucket, _o, _a, _r = _setup(2, 32, D, A, hidden, layers, seed=2, cont=cont)
    lay = H.mlp_layout(pol, bucket)
    assert lay["wide"]
    g = torch.Generator(device="cuda").manual_seed(N)
    obs = torch.randn(N, D, device="cuda", generator=g)
    with torch.no_grad():
        v_ref = pol.value(obs)
        if cont:
            noise = torch.randn(N, A, device="cuda", generator=g)
            mean = pol.actor(obs)
            std = pol.actor_logstd.exp().expand_as(mean)
            a_ref = mean + std * noise
            lp_ref = torch.distributions.Normal(mean, std).log_prob(a_ref).sum(1)
        else:
            noise = torch.rand(N, device="cuda", generator=g)
            logits = pol.actor(obs)
            cdf = torch.softmax(logits, 1).cumsum(1)
            a_ref = (noise[:, None] >= cdf).sum(1).clamp(max=A - 1).float()
            lp_ref = torch.log_softmax(logits, 1).gather(1, a_ref.long()[:, None])[:, 0]
    a, lp, v = H.mlp_act(obs, noise, bucket.flat_param, lay)
    # 1e-5 absolute (north_star): a value is a 128-term fp32 sum of O(1) terms here, summed in a different order than rocBLAS does
    np.testing.assert_allclose(v.cpu().numpy(), v_ref.cpu().numpy(), rtol=2e-5, atol=1e-5)
    if cont:
        np.testing.assert_allclose(a.cpu().numpy(), a_ref.cpu().numpy(), rtol=2e-5, atol=1e-5)
        np.testing.assert_allclose(lp.cpu().numpy(), lp_ref.cpu().numpy(), rtol=2e-5, atol=2e-5)
    else:
        # a draw within rounding of a CDF boundary may land on either side
        same = (a == a_ref)
        assert float(same.float().mean()) >= 0.99
        np.testing.assert_allclose(lp[same].cpu().numpy(), lp_ref[same].cpu().numpy(), rtol=2e-5, atol=2e-5)
    _, _, v2 = H.mlp_act(obs, None, bucket.flat_param, lay)                  # value only (the bootstrap)
    assert torch.equal(v2, v)


def _hp(N, T, Dm, A, **kw):
    hp = dict(gym_id="Synthetic-v0", seed=1.0, num_steps=T, gae=True, total_timesteps=T * N, anneal_lr=False,
              gae_lambda=0.95, num_update_epochs=4, num_envs=N, num_minibatches=4, entropy_coeff=0.01,
              value_coeff=0.5, clip_coeff=0.2, clip_vloss=True, max_grad_norm=0.5, target_kl=None, norm_adv=True,
              capture_video=False, hidden_dim=64, continuous=True, learning_rate=3e-4, exp_name="t", num_layers=2,
              dropout=0.0, gamma=0.99, track=False, log=False, save=False, obs_dim=Dm, act_dim=A)
    hp.update(kw)
    return hp


@pytest.mark.parametrize("launch", ["eager", "hipGraph"])
@pytest.mark.parametrize("hidden,layers,Dm", [(128, 2, 64), (64, 3, 24), (128, 3, 128), (32, 1, 8)])
def test_full_update_with_a_wide_policy_matches_oracle(hidden, layers, Dm, launch):
    """One whole ``ppo.update`` (GAE, E x minibatches of K7w + K6b) on the SURVEY 8d tensors against
    ``oracle.reference_update`` with the same net shape: permutations bit-exact, advantages 1e-5, every step's loss
    scalars, the final weights -- the treatment the default shape gets in test_parity_fullsize.py."""
    import bench
    from aur_ppo_amd.ppo import ppo
    from oracle import ppo_oracle as O
    T, N, A = 32, 256, 6
    hp = _hp(N, T, Dm, A, hidden_dim=hidden, num_layers=layers, hip_graph=(launch == "hipGraph"))
    torch.manual_seed(1)
    agent = ppo(hp)
    assert agent._mlp is not None and agent._mlp["wide"] and agent._fused_adam
    data = bench.synth_buffers(T, N, Dm, A, 1234)
    init_sd = {k: v.detach().cpu().clone() for k, v in agent.policy.state_dict().items()}
    for k in ("states", "actions", "values", "rewards", "terminals"):
        getattr(agent.buffer, k).copy_(data[k])
    with torch.no_grad():
        _, lp, _, _ = agent.policy.evaluate(agent.buffer.states.view(-1, Dm), agent.buffer.actions.view(-1, A))
        agent.buffer.log_probs.copy_(lp.view(T, N))
    data["log_probs"] = agent.buffer.log_probs.cpu()
    agent.seed_all(1)
    if launch == "hipGraph":
        agent._graph_state = 1
    ret, adv = agent.advantages(data["next_obs"].cuda(), data["next_done"].cuda())
    n = agent.update(ret, adv)
    torch.cuda.synchronize()
    assert (agent._graph is not None) == (launch == "hipGraph") and n == 16
    net = O.make_actor_critic(Dm, (A,), hidden, layers, True)
    net.load_state_dict(init_sd)
    opt = torch.optim.Adam(net.parameters(), lr=hp["learning_rate"], eps=1e-5)
    buf = {k: data[k] for k in ("states", "actions", "log_probs", "rewards", "terminals", "values")}
    res = O.reference_update(net, opt, buf, data["next_obs"], data["next_done"], hp, np.random.RandomState(1))
    perms = agent._last_perms.cpu().numpy()
    for e in range(4):
        assert np.array_equal(perms[e], res["perms"][e]), f"epoch {e} permutation"
    np.testing.assert_allclose(adv.cpu().numpy(), res["advantages"].numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), res["returns"].numpy(), rtol=0, atol=1e-5)
    got = agent._scalars[:n].cpu().numpy()
    cols = [0, 1, 2, 3, 4, 5, 7, 8]
    np.testing.assert_allclose(got[:, cols], res["scalars"][:, cols], rtol=1e-4, atol=1e-5)
    assert np.abs(got[:, 6] - res["scalars"][:, 6]).max() <= 1.5 / agent.minibatch_size
    # 2e-5 absolute: tiles reach workgroups through a counter, a gradient's last bits depend on the launch, and Adam's step
    # has sensitivity lr / eps = 30 to an element far below eps (tools/update_repeatability.py: two GPU runs of one update
    # end up to 9e-6 apart at the bench size; bound 16 steps x lr x 4e-7 max|g| / eps)
    for k, v in agent.policy.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), net.state_dict()[k].numpy(), rtol=1e-4, atol=2e-5, err_msg=k)


def test_rollout_with_a_wide_policy_goes_through_k8w_and_trains():
    """ppo.train() with -d 128 -nl 3: the rollout step is K8w (also inside the captured rollout graph), the update K7w."""
    from aur_ppo_amd.ppo import ppo
    hp = _hp(64, 16, 24, 4, hidden_dim=128, num_layers=3, total_timesteps=4 * 16 * 64)
    torch.manual_seed(2)
    a = ppo(hp)
    assert a._mlp["wide"]
    a.train()
    torch.cuda.synchronize()
    assert a.last_update["scalars"].shape == (16, 9) and np.isfinite(a.last_update["scalars"]).all()
    assert torch.isfinite(a.bucket.flat_param).all()


@pytest.mark.parametrize("handover", [False, True], ids=["prepare-each-call", "handed-over"])
@pytest.mark.parametrize("hidden,layers,D", [(128, 3, 64), (64, 3, 24), (32, 1, 8), (96, 2, 100)])
def test_wide_minibatch_matches_step_then_clip_adam(hidden, layers, D, handover, monkeypatch):
    """aurppo_mlp_wide_ppo_minibatch_f32 over a run of minibatches == the same run as aurppo_mlp_wide_ppo_step_f32 +
    aurppo_clip_adam_f32 pairs: parameters, moments, loss scalars, norms and the Adam step count.  Tiles are dealt by
    static stride in both runs, so the two see the same summation order (tests/test_determinism.py) and the norms can be held
    to 1e-6 over eight Adam steps at lr 3e-3."""
    monkeypatch.setenv("AURPPO_STATIC_TILES", "1")
    T, N, A, M = 16, 64, 6, 300      # B = 1024: three full slices and a ragged tail
    H, pol, bucket, obs, act, rec = _setup(T, N, D, A, hidden, layers, seed=3)
    lay = H.mlp_layout(pol, bucket)
    nb = bucket.flat_param.numel()
    perm = torch.randperm(T * N, device="cuda").int()
    slices = [perm[s:s + M] for s in range(0, T * N, M)]
    p0 = bucket.flat_param.clone()

    def run(chained):
        bucket.flat_param.copy_(p0)
        m, v = torch.zeros(nb, device="cuda"), torch.zeros(nb, device="cuda")
        lr, t = torch.full((1,), 3e-3, device="cuda"), torch.zeros(1, device="cuda")
        sc = torch.zeros(len(slices) * 2, 9, device="cuda")
        norms = torch.zeros(len(slices) * 2, device="cuda")
        g = torch.zeros(nb, device="cuda")
        k = 0
        seq = slices + slices
        for _rep in range(2):
            for idx in slices:
                if chained:
                    # handed-over: every call names the slice stepped next (the optimizer launch refreshes the operand-order
                    # copies and forms that slice's statistics) and all but the first skip their prepare launch
                    nxt = seq[k + 1] if (handover and k + 1 < len(seq)) else None
                    H.mlp_ppo_minibatch(obs, act, rec, idx, bucket.flat_param, lay, g, 0.2, 0.01, 0.5, True, 1, sc[k], m, v, lr, t,
                                        0.5, (0.9, 0.999), 1e-5, norms[k:k + 1], next_idx=nxt, chained=handover and k > 0)
                else:
                    H.mlp_ppo_step(obs, act, rec, idx, bucket.flat_param, lay, g, 0.2, 0.01, 0.5, True, 1, sc[k])
                    H.clip_adam_(bucket.flat_param, g, m, v, lr, t, 0.5, None, (0.9, 0.999), 1e-5, norms[k:k + 1])
                k += 1
        torch.cuda.synchronize()
        return bucket.flat_param.clone(), m, v, sc, norms, float(t)

    ref = run(False)
    got = run(True)
    assert got[5] == ref[5] == 2 * len(slices)
    torch.testing.assert_close(got[4], ref[4], rtol=1e-6, atol=0)
    torch.testing.assert_close(got[3], ref[3], rtol=2e-5, atol=2e-6)
    for a_, b_ in zip(got[:3], ref[:3]):
        torch.testing.assert_close(a_, b_, rtol=1e-4, atol=1e-6)
    assert float((ref[0] - p0).abs().max()) > 1e-3
