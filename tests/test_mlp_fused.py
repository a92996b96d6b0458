"""GPU: K7 (fused gather + MLP forward + PPO loss + backward) vs the per-op autograd path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(T, N, D, A, seed=0, cont=True):
    from aur_ppo_amd import hip_ops as H
    from aur_ppo_amd.actor_critic import actor_critic
    from aur_ppo_amd.flat import FlatBucket
    torch.manual_seed(seed)
    pol = actor_critic(D, (A,) if cont else A, 64, 2, 0.0, cont).cuda()
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


@pytest.fixture(params=["2", "3"], ids=["f32-mfma", "bf16x3-mfma"])
def variant(request, monkeypatch):
    """The builds of K7: k_mlp_step3 (bf16 MFMA over exact three-way splits, the default) and k_mlp_step2 (fp32 MFMA, the plain-fp32
    reference point, AURPPO_K7_VARIANT=2), held to the SAME tolerances."""
    monkeypatch.setenv("AURPPO_K7_VARIANT", request.param)
    return request.param


@pytest.mark.parametrize("T,N,D,A,M", [(8, 64, 64, 6, 256), (16, 64, 64, 6, 1000), (4, 32, 4, 1, 77), (8, 32, 32, 16, 128),
                                         (8, 32, 6, 3, 100), (8, 64, 10, 9, 31), (8, 32, 3, 1, 100), (8, 64, 11, 3, 333),
                                         (16, 64, 17, 6, 1000), (8, 32, 16, 8, 100), (128, 1024, 64, 6, 131072)])
@pytest.mark.parametrize("norm_adv,vmode", [(True, 1), (False, 2), (True, 0)])
def test_fused_step_matches_autograd_path(T, N, D, A, M, norm_adv, vmode, variant):
    H, pol, bucket, obs, act, rec = _setup(T, N, D, A)
    B = T * N
    idx = torch.randperm(B, device="cuda")[:M].int()
    lay = H.mlp_layout(pol, bucket)
    assert lay is not None and lay["D"] == D and lay["A"] == A
    # reference: the existing per-op path (K3 gather -> torch evaluate -> K5 loss -> autograd)
    mb = H.gather(idx, [obs, act, rec])
    _, nlp, ent, nv = pol.evaluate(mb[0], mb[1])
    sc_ref = torch.empty(9, device="cuda")
    loss = H.ppo_loss_packed(nlp, nv, ent, mb[2], 0.2, 0.01, 0.5, norm_adv, vmode, sc_ref)
    bucket.zero_grad()
    loss.backward()
    g_ref = bucket.flat_grad[:lay["n_params"]].clone()
    # K7
    g_out = torch.full_like(bucket.flat_grad, float("nan"))
    sc = H.mlp_ppo_step(obs, act, rec, idx, bucket.flat_param, lay, g_out, 0.2, 0.01, 0.5, norm_adv, vmode)
    torch.cuda.synchronize()
    np.testing.assert_allclose(sc.cpu().numpy(), sc_ref.cpu().numpy(), rtol=2e-5, atol=2e-6)
    g = g_out[:lay["n_params"]]
    assert torch.isfinite(g).all()
    scale = float(g_ref.abs().max())
    err = float((g - g_ref).abs().max())
    assert err <= 2e-5 * scale + 1e-8, (err, scale)
    # per-parameter relative check where the gradient is not tiny
    names = ["actor_logstd"] + [f"{n}.{k}" for n in ("actor", "critic") for k in ("w1", "b1", "w2", "b2", "w3", "b3")]
    off = 0
    for p_, nm in zip(bucket.params, names):
        k = p_.numel()
        a, b = g[off:off + k], g_ref[off:off + k]
        s = float(b.abs().max())
        # floor: a bias gradient is a sum of M terms that may cancel to far below the terms' own rounding error
        assert float((a - b).abs().max()) <= 5e-5 * s + 2e-6 * scale + 1e-9, (nm, float((a - b).abs().max()), s)
        off += k


def test_fused_step_rejects_unsupported_shapes():
    H, pol, bucket, obs, act, rec = _setup(4, 32, 64, 6)
    lay = dict(H.mlp_layout(pol, bucket))
    idx = torch.arange(16, device="cuda", dtype=torch.int32)
    lay["D"] = 63
    with pytest.raises((RuntimeError, ValueError)):
        H.mlp_ppo_step(obs, act, rec, idx, bucket.flat_param, lay, bucket.flat_grad, 0.2, 0.0, 0.5)
    from aur_ppo_amd.actor_critic import actor_critic
    from aur_ppo_amd.flat import FlatBucket
    wide = actor_critic(64, (6,), 128, 2, 0.0, True).cuda()
    wl = H.mlp_layout(wide, FlatBucket(wide.parameters()))
    assert wl is not None and wl["wide"] and wl["hidden"] == 128          # K7w / K8w (tests/test_mlp_wide.py)
    for shape in ((64, (6,), 256, 2), (64, (6,), 64, 4), (129, (6,), 128, 2), (8, (17,), 64, 2)):
        pol2 = actor_critic(shape[0], shape[1], shape[2], shape[3], 0.0, True).cuda()
        assert H.mlp_layout(pol2, FlatBucket(pol2.parameters())) is None, shape
    with pytest.raises(ValueError):       # the chained minibatch kernels are the 64-64 shape's alone
        H.mlp_ppo_grad(obs, act, rec, idx, bucket.flat_param, wl, bucket.flat_grad, 0.2, 0.0, 0.5, True, 1,
                       torch.empty(9, device="cuda"), torch.zeros(1, device="cuda"))


@pytest.mark.parametrize("T,N,D,A,M", [(8, 64, 4, 2, 200), (16, 64, 64, 16, 1024), (8, 64, 6, 11, 333), (8, 64, 5, 3, 200),
                                         (128, 256, 8, 5, 32768)])
@pytest.mark.parametrize("norm_adv,vmode,ec", [(True, 1, 0.01), (False, 2, 0.05)])
def test_fused_step_categorical_head_matches_autograd_path(T, N, D, A, M, norm_adv, vmode, ec, variant):
    """Discrete policy (CartPole-style, BASELINE configs[0]): Categorical log-prob / entropy and their gradients."""
    H, pol, bucket, obs, act, rec = _setup(T, N, D, A, seed=1, cont=False)
    idx = torch.randperm(T * N, device="cuda")[:M].int()
    lay = H.mlp_layout(pol, bucket)
    assert lay is not None and lay["continuous"] is False and lay["A"] == A
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
    for p_ in bucket.params:      # per tensor: relative to its own scale, with a floor at the global rounding level
        k = p_.numel()
        a, b = g[off:off + k], g_ref[off:off + k]
        s_ = float(b.abs().max())
        assert float((a - b).abs().max()) <= 5e-5 * s_ + 2e-6 * gscale + 1e-9, (off, float((a - b).abs().max()), s_)
        off += k


# ---------------------------------------------------------------------------------- K8: rollout step
@pytest.mark.parametrize("N,D,A,cont", [(4096, 64, 6, True), (77, 4, 2, False), (256, 16, 16, True), (33, 8, 5, False), (1, 64, 1, True),
                                        (100, 11, 3, True), (64, 3, 4, False)])
def test_act_kernel_matches_torch_formulas(N, D, A, cont):
    H, pol, bucket, _obs, _act, _rec = _setup(2, 32, D, A, seed=2, cont=cont)
    lay = H.mlp_layout(pol, bucket)
    g = torch.Generator(device="cuda").manual_seed(N)
    obs = torch.randn(N, D, device="cuda", generator=g)
    with torch.no_grad():
        v_ref = pol.value(obs)
        if cont:
            noise = torch.randn(N, A, device="cuda", generator=g)
            mean = pol.actor(obs)
            std = pol.actor_logstd.exp().expand_as(mean)
            a_ref = mean + std * noise
            _, lp_ref, _, _ = pol.evaluate(obs, a_ref)
        else:
            noise = torch.rand(N, device="cuda", generator=g)
            logp_all = torch.log_softmax(pol.actor(obs), -1)
            cdf = logp_all.exp().cumsum(-1)
            a_ref = (noise.unsqueeze(1) >= cdf).sum(1).clamp(max=A - 1)
            lp_ref = logp_all.gather(1, a_ref.unsqueeze(1)).squeeze(1)
    actions, logp, value = H.mlp_act(obs, noise, bucket.flat_param, lay)
    torch.testing.assert_close(value, v_ref, rtol=1e-5, atol=2e-6)
    if cont:
        torch.testing.assert_close(actions, a_ref, rtol=1e-5, atol=2e-6)
        torch.testing.assert_close(logp, lp_ref, rtol=1e-5, atol=2e-5)
    else:
        same = actions.long() == a_ref
        assert same.float().mean() > 0.99      # a draw within rounding of a CDF edge may fall either side
        torch.testing.assert_close(logp[same], lp_ref[same], rtol=1e-5, atol=2e-6)
    # value-only mode (the bootstrap) and writing straight into buffer rows
    _, _, v2 = H.mlp_act(obs, None, bucket.flat_param, lay)
    assert torch.equal(v2, value)
    buf_a = torch.zeros((3, N, A) if cont else (3, N), device="cuda")
    buf_lp, buf_v = torch.zeros(3, N, device="cuda"), torch.zeros(3, N, device="cuda")
    H.mlp_act(obs, noise, bucket.flat_param, lay, buf_a[1], buf_lp[1], buf_v[1])
    assert torch.equal(buf_a[1], actions) and torch.equal(buf_lp[1], logp) and torch.equal(buf_v[1], value)
    assert float(buf_a[0].abs().sum() + buf_a[2].abs().sum() + buf_v[0].abs().sum() + buf_lp[2].abs().sum()) == 0.0


# ---------------------------------------------------------------------------------- K7 + K6b chained
@pytest.mark.parametrize("cont", [True, False])
def test_chained_minibatches_match_step_then_clip_adam(cont, variant):
    """aurppo_mlp_ppo_minibatch_f32 over a run of minibatches (each call preparing the next) == the same run as
    aurppo_mlp_ppo_step_f32 + aurppo_clip_adam_f32 pairs: parameters, moments, loss scalars and norms."""
    T, N, D, A, M = 16, 64, 64 if cont else 6, 6 if cont else 5, 300      # B = 1024: three full slices and a ragged tail
    H, pol, bucket, obs, act, rec = _setup(T, N, D, A, seed=3, cont=cont)
    lay = H.mlp_layout(pol, bucket)
    n = lay["n_params"]
    nb = bucket.flat_param.numel()      # the bucket may carry alignment padding past the policy
    assert nb >= n
    perm = torch.randperm(T * N, device="cuda").int()
    slices = [perm[s:s + M] for s in range(0, T * N, M)]
    assert slices[-1].numel() not in (0, M)
    p0 = bucket.flat_param.clone()

    def run(chained):
        bucket.flat_param.copy_(p0)
        m, v = torch.zeros(nb, device="cuda"), torch.zeros(nb, device="cuda")
        lr, t = torch.full((1,), 3e-3, device="cuda"), torch.zeros(1, device="cuda")
        sc = torch.zeros(len(slices) * 2, 9, device="cuda")
        norms = torch.zeros(len(slices) * 2, device="cuda")
        g = torch.zeros(nb, device="cuda")
        k = 0
        for rep in range(2):
            for i, idx in enumerate(slices):
                if chained:
                    nxt = slices[i + 1] if i + 1 < len(slices) else (slices[0] if rep == 0 else None)
                    H.mlp_ppo_minibatch(obs, act, rec, idx, bucket.flat_param, lay, g, 0.2, 0.01, 0.5, True, 1, sc[k], m, v, lr, t,
                                        0.5, (0.9, 0.999), 1e-5, norms[k:k + 1], next_idx=nxt, chained=k > 0)
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
    assert float((ref[0] - p0).abs().max()) > 1e-3      # the run did move the parameters


def test_grad_and_apply_halves_match_the_whole_minibatch(variant):
    """aurppo_mlp_ppo_grad_f32 -> (an all-reduce's SUM, emulated by doubling) -> aurppo_mlp_ppo_apply_f32(grad_scale = 1/2)
    over a run of chained minibatches == aurppo_mlp_ppo_minibatch_f32 over the same run."""
    T, N, D, A, M = 16, 64, 64, 6, 300
    H, pol, bucket, obs, act, rec = _setup(T, N, D, A, seed=4, cont=True)
    lay = H.mlp_layout(pol, bucket)
    n, nb = lay["n_params"], bucket.flat_param.numel()
    perm = torch.randperm(T * N, device="cuda").int()
    slices = [perm[s:s + M] for s in range(0, T * N, M)]
    p0 = bucket.flat_param.clone()

    def run(split):
        bucket.flat_param.copy_(p0)
        m, v, g = (torch.zeros(nb, device="cuda") for _ in range(3))
        lr, t = torch.full((1,), 3e-3, device="cuda"), torch.zeros(1, device="cuda")
        sc, norms = torch.zeros(len(slices), 9, device="cuda"), torch.zeros(len(slices), device="cuda")
        for k, idx in enumerate(slices):
            nxt = slices[k + 1] if k + 1 < len(slices) else None
            if split:
                H.mlp_ppo_grad(obs, act, rec, idx, bucket.flat_param, lay, g, 0.2, 0.01, 0.5, True, 1, sc[k], t, chained=k > 0)
                g.mul_(2.0)                                   # what a 2-rank SUM of identical shards would leave
                H.mlp_ppo_apply(bucket.flat_param, g, m, v, lay, lr, t, 0.5, (0.9, 0.999), 1e-5, norms[k:k + 1], grad_scale=0.5,
                                rec=rec, next_idx=nxt)
            else:
                H.mlp_ppo_minibatch(obs, act, rec, idx, bucket.flat_param, lay, g, 0.2, 0.01, 0.5, True, 1, sc[k], m, v, lr, t, 0.5,
                                    (0.9, 0.999), 1e-5, norms[k:k + 1], next_idx=nxt, chained=k > 0)
        torch.cuda.synchronize()
        return bucket.flat_param.clone(), m, v, sc, norms, float(t)

    ref, got = run(False), run(True)
    assert got[5] == ref[5] == len(slices)
    torch.testing.assert_close(got[4], ref[4], rtol=1e-6, atol=0)
    torch.testing.assert_close(got[3], ref[3], rtol=1e-6, atol=1e-7)
    for a_, b_ in zip(got[:3], ref[:3]):
        torch.testing.assert_close(a_, b_, rtol=1e-5, atol=1e-7)


# ---------------------------------------------------------------------------------- packed 64-byte records
@pytest.mark.parametrize("cont,D,A", [(True, 64, 6), (True, 11, 12), (False, 8, 5)])
def test_packed_records_give_identical_results(cont, D, A, variant):
    """actions == NULL + (B, 16) records from aurppo_pack_records_f32: same gradients and scalars, bit for bit, as the
    separate action buffer -- through the plain step, the chained minibatch and the two halves."""
    T, N, M = 16, 64, 300
    H, pol, bucket, obs, act, rec = _setup(T, N, D, A, seed=6, cont=cont)
    lay = H.mlp_layout(pol, bucket)
    nb = bucket.flat_param.numel()
    rec64 = H.pack_records(rec, act.reshape(T * N, -1))
    assert rec64.shape == (T * N, 16) and torch.equal(rec64[:, :4], rec)
    aw = A if cont else 1
    assert torch.equal(rec64[:, 4:4 + aw], act.reshape(T * N, -1)) and float(rec64[:, 4 + aw:].abs().sum()) == 0.0
    perm = torch.randperm(T * N, device="cuda").int()
    slices = [perm[s:s + M] for s in range(0, T * N, M)]
    p0 = bucket.flat_param.clone()

    def run(packed, mode):
        bucket.flat_param.copy_(p0)
        a_, r_ = (None, rec64) if packed else (act, rec)
        m, v, g = (torch.zeros(nb, device="cuda") for _ in range(3))
        lr, t = torch.full((1,), 3e-3, device="cuda"), torch.zeros(1, device="cuda")
        sc, norms = torch.zeros(len(slices), 9, device="cuda"), torch.zeros(len(slices), device="cuda")
        grads = []
        for k, idx in enumerate(slices):
            nxt = slices[k + 1] if k + 1 < len(slices) else None
            if mode == "step":
                H.mlp_ppo_step(obs, a_, r_, idx, bucket.flat_param, lay, g, 0.2, 0.01, 0.5, True, 1, sc[k])
                grads.append(g.clone())
            elif mode == "chain":
                H.mlp_ppo_minibatch(obs, a_, r_, idx, bucket.flat_param, lay, g, 0.2, 0.01, 0.5, True, 1, sc[k], m, v, lr, t, 0.5,
                                    (0.9, 0.999), 1e-5, norms[k:k + 1], next_idx=nxt, chained=k > 0)
            else:
                H.mlp_ppo_grad(obs, a_, r_, idx, bucket.flat_param, lay, g, 0.2, 0.01, 0.5, True, 1, sc[k], t, chained=k > 0)
                H.mlp_ppo_apply(bucket.flat_param, g, m, v, lay, lr, t, 0.5, (0.9, 0.999), 1e-5, norms[k:k + 1], rec=r_, next_idx=nxt)
        torch.cuda.synchronize()
        return bucket.flat_param.clone(), sc, norms, grads

    for mode in ("step", "chain", "halves"):
        ref, got = run(False, mode), run(True, mode)
        assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]) and torch.equal(got[0], ref[0]), mode
        for a_, b_ in zip(got[3], ref[3]):
            assert torch.equal(a_, b_)
