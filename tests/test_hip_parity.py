"""GPU parity: the HIP path (through the C ABI) vs the oracle and the golden vectors."""
import hashlib

import numpy as np
import pytest
import torch

from tests.util import gae_big_inputs, load, synth_rollout

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from aur_ppo_amd import hip_ops
    assert torch.cuda.is_available()
    return hip_ops


@pytest.fixture(scope="module")
def O():
    from oracle import ppo_oracle
    return ppo_oracle


@pytest.fixture(scope="module")
def CO():
    from oracle import c_oracle
    return c_oracle


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


# ---------------------------------------------------------------------------------- K1
def test_gae_golden_bit_exact(H):
    z = load("gae.npz")
    n = 0
    for name in z["names"]:
        T, N, gamma, lam = z[f"{name}/meta"]
        T, N = int(T), int(N)
        big = f"{name}/seed" in z.files
        if big:
            args = gae_big_inputs(int(z[f"{name}/seed"][1]), T, N)
        else:
            args = [z[f"{name}/{k}"] for k in ("rewards", "values", "terminals", "next_value", "next_done")]
        targs = [dev(a) for a in args]
        for mode, key in ((H.GAE, "gae"), (H.NORMAL_ADV, "norm"), (H.GAE_SKIP_LAST, "skip")):
            if big:
                if key == "skip":
                    continue
                ret, adv = H.gae(*targs, gamma, lam, mode)
                assert (_sha(adv.cpu().numpy()) == z[f"{name}/adv_{key}_sha"]).all(), (name, key)
                assert (_sha(ret.cpu().numpy()) == z[f"{name}/ret_{key}_sha"]).all(), (name, key)
            elif f"{name}/adv_{key}" in z.files:
                ret, adv = H.gae(*targs, gamma, lam, mode)
                np.testing.assert_array_equal(adv.cpu().numpy(), z[f"{name}/adv_{key}"], err_msg=f"{name} {key}")
                np.testing.assert_array_equal(ret.cpu().numpy(), z[f"{name}/ret_{key}"], err_msg=f"{name} {key}")
            n += 1
    assert n > 40


@pytest.mark.parametrize("T,N", [(1, 1), (3, 17), (128, 100), (129, 16), (300, 33), (2048, 1), (1024, 5), (64, 8192),
                                 (8, 16400)])
def test_gae_ragged_shapes_vs_oracle(H, CO, T, N):
    rs = np.random.RandomState(T * 131 + N)
    r, v = rs.standard_normal((T, N)).astype(np.float32), rs.standard_normal((T, N)).astype(np.float32)
    d = (rs.random_sample((T, N)) < 0.05).astype(np.float32)
    nv, nd = rs.standard_normal(N).astype(np.float32), (rs.random_sample(N) < 0.3).astype(np.float32)
    for mode in (0, 1, 2):
        ret, adv = H.gae(dev(r), dev(v), dev(d), dev(nv), dev(nd), 0.99, 0.95, mode)
        ret_o, adv_o = CO.gae(r, v, d, nv, nd, 0.99, 0.95, mode)
        np.testing.assert_array_equal(adv.cpu().numpy(), adv_o)
        np.testing.assert_array_equal(ret.cpu().numpy(), ret_o)


def test_gae_properties_at_full_size(H):
    # BASELINE size N=4096,T=128: linearity in rewards when values=0 and no terminals; lam=1 telescopes
    T, N = 128, 4096
    rs = np.random.RandomState(5)
    r1, r2 = (dev(rs.standard_normal((T, N)).astype(np.float32)) for _ in range(2))
    zero, zN = torch.zeros(T, N, device="cuda"), torch.zeros(N, device="cuda")
    _, a1 = H.gae(r1, zero, zero, zN, zN, 0.99, 0.95)
    _, a2 = H.gae(r2, zero, zero, zN, zN, 0.99, 0.95)
    _, a12 = H.gae(r1 + r2, zero, zero, zN, zN, 0.99, 0.95)
    torch.testing.assert_close(a12, a1 + a2, rtol=1e-5, atol=1e-5)
    v = dev(rs.standard_normal((T, N)).astype(np.float32))
    d = dev((rs.random_sample((T, N)) < 0.02).astype(np.float32))
    nv = dev(rs.standard_normal(N).astype(np.float32))
    ret_g, adv_g = H.gae(r1, v, d, nv, zN, 0.99, 1.0, H.GAE)
    ret_n, adv_n = H.gae(r1, v, d, nv, zN, 0.99, 1.0, H.NORMAL_ADV)
    torch.testing.assert_close(adv_g, adv_n, rtol=1e-4, atol=1e-4)
    # all-terminal mask: A = r - V exactly
    one = torch.ones(T, N, device="cuda")
    _, a = H.gae(r1, v, one, nv, torch.ones(N, device="cuda"), 0.99, 0.95)
    assert torch.equal(a, r1 - v)


def test_gae_rejects_bad_arguments(H):
    x = torch.zeros(4, 4, device="cuda")
    n = torch.zeros(4, device="cuda")
    with pytest.raises(ValueError):
        H.gae(x, x[:2], x, n, n, 0.99, 0.95)
    with pytest.raises(RuntimeError, match="bad mode"):
        H.gae(x, x, x, n, n, 0.99, 0.95, mode=7)
    with pytest.raises(ValueError):
        H.gae(x.cpu(), x, x, n, n, 0.99, 0.95)


# ---------------------------------------------------------------------------------- K2
@pytest.mark.parametrize("B", [8, 16, 512, 4096])
def test_shuffle_golden_small_bit_exact(H, B):
    z = load("shuffle.npz")
    rng = H.MT19937(1, B)
    got = torch.cat([rng.shuffle_epochs(B, 4), rng.shuffle_epochs(B, 4)]).cpu().numpy()
    np.testing.assert_array_equal(got, z[f"B{B}/perms"])
    key, pos = rng.get_state()
    np.testing.assert_array_equal(key, z[f"B{B}/state_key"])
    assert pos == int(z[f"B{B}/state_pos"][0])


@pytest.mark.parametrize("B", [65536, 131072, 524288])
def test_shuffle_golden_large_digests(H, B):
    z = load("shuffle.npz")
    rng = H.MT19937(1, B)
    got = torch.cat([rng.shuffle_epochs(B, 4), rng.shuffle_epochs(B, 4)]).cpu().numpy()
    for k in range(8):
        assert (_sha(got[k]) == z[f"B{B}/sha"][k]).all(), (B, k)
        np.testing.assert_array_equal(got[k][:16], z[f"B{B}/head"][k])
        np.testing.assert_array_equal(got[k][-16:], z[f"B{B}/tail"][k])
        assert np.array_equal(np.sort(got[k]), np.arange(B))   # bijection
    key, pos = rng.get_state()
    np.testing.assert_array_equal(key, z[f"B{B}/state_key"])
    assert pos == int(z[f"B{B}/state_pos"][0])


@pytest.mark.parametrize("seed,n", [(0, 2), (1, 3), (7, 1000), (12345, 625), (2**32 - 1, 1249), (99, 70000), (3, 1), (3, 0)])
def test_shuffle_inplace_matches_numpy_any_seed(H, seed, n):
    rs = np.random.RandomState(seed)
    rng = H.MT19937(seed, max(n, 1))
    x = np.arange(n)[::-1].copy()
    idx = dev(x.astype(np.int32))
    for _ in range(3):   # stream continues across calls
        rs.shuffle(x)
        rng.shuffle_(idx)
        np.testing.assert_array_equal(idx.cpu().numpy(), x)
    st = rs.get_state()
    key, pos = rng.get_state()
    np.testing.assert_array_equal(key, st[1])
    assert pos == st[2]


@pytest.mark.parametrize("accept,wgs", [("1", "6"), ("3", "1"), ("3", "2"), ("3", "6"), ("3", "8")])
def test_shuffle_every_accept_kernel_matches_numpy(H, accept, wgs, monkeypatch):
    """The two builds of the accept stage (k_fy_accept: one workgroup, rounds; k_fy_accept3: relay between workgroups, the default) and the relay's widths, over sizes from below one 16 384-draw chunk to dozens of them, the stream carried
    across consecutive shuffles: permutations and the generator state afterwards equal numpy's."""
    monkeypatch.setenv("AURPPO_K2_ACCEPT", accept)
    monkeypatch.setenv("AURPPO_K2_ACCEPT3_WGS", wgs)
    for n in (2, 11, 1000, 11500, 12000, 23500, 40000, 70000, 300000, 524288):
        rs = np.random.RandomState(7)
        rng = H.MT19937(7, n)
        got = rng.shuffle_epochs(n, 3).cpu().numpy()
        idx = np.arange(n)
        for e in range(3):
            rs.shuffle(idx)
            np.testing.assert_array_equal(got[e], idx, err_msg=f"accept {accept}, {wgs} workgroups, n={n}, shuffle {e}")
        key, pos = rng.get_state()
        st = rs.get_state()
        np.testing.assert_array_equal(key, st[1])
        assert pos == st[2]
        status = torch.zeros(1, device="cuda")
        rng.status_into(status)
        assert float(status) == 0.0


@pytest.mark.parametrize("accept", ["1", "3"])
def test_shuffle_that_runs_out_of_draws_raises_the_sticky_flag_and_reseeding_recovers(H, accept, monkeypatch):
    """np.random.shuffle cannot fail; the device twin works from pre-generated draws (expectation + 12 sigma) and says so if a
    shuffle ever needs more: every accept kernel must then stop (no hang: the relay's workgroups leave on the `done` word), raise
    the flag aurppo_mt19937_status_f32 reports, and a reseed must clear it.  AURPPO_TEST_K2_STARVE makes the twist keep 60 % of
    one shuffle's draws in stock."""
    monkeypatch.setenv("AURPPO_K2_ACCEPT", accept)
    n = 200000
    monkeypatch.setenv("AURPPO_TEST_K2_STARVE", "60")
    rng = H.MT19937(3, n)
    rng.shuffle_epochs(n, 2)
    status = torch.zeros(1, device="cuda")
    rng.status_into(status)
    torch.cuda.synchronize()
    assert float(status) == 1.0
    monkeypatch.delenv("AURPPO_TEST_K2_STARVE")
    rng.seed(3)
    got = rng.shuffle_epochs(n, 1).cpu().numpy()[0]
    rs = np.random.RandomState(3)
    idx = np.arange(n)
    rs.shuffle(idx)
    np.testing.assert_array_equal(got, idx)
    rng.status_into(status)
    assert float(status) == 0.0


def test_shuffle_state_roundtrip_and_reseed(H):
    rng = H.MT19937(5, 1000)
    a = rng.shuffle_epochs(1000, 1).clone()
    key, pos = rng.get_state()
    b = rng.shuffle_epochs(1000, 1).clone()
    rng.set_state(key, pos)
    b2 = rng.shuffle_epochs(1000, 1)
    assert torch.equal(b, b2)
    rng.seed(5)
    assert torch.equal(rng.shuffle_epochs(1000, 1), a)
    with pytest.raises(RuntimeError, match="max_n"):
        rng.shuffle_epochs(1001, 1)


# ---------------------------------------------------------------------------------- K3
@pytest.mark.parametrize("M,rows", [(1, [64]), (1000, [64, 6, 1, 1, 1, 1]), (4096, [4, 2, 1]), (257, [5, 3, 7]),
                                    (33, [128 * 128, 1, 5]), (2048, [256]), (16384, [64, 6, 1, 1, 1, 1])])
def test_gather_bit_exact(H, M, rows):
    B = max(2 * M, 64)
    rs = np.random.RandomState(M)
    srcs = [dev(rs.standard_normal((B, r) if r > 1 else (B,)).astype(np.float32)) for r in rows]
    idx_np = rs.randint(0, B, size=M).astype(np.int32)
    idx = dev(idx_np)
    outs = H.gather(idx, srcs)
    for o, s in zip(outs, srcs):
        assert torch.equal(o, s[idx.long()])


def test_gather_unaligned_views_and_errors(H):
    base = torch.randn(1000 * 6 + 1, device="cuda")
    src = base[1:].view(1000, 6)            # 4-byte aligned only -> scalar path
    idx = torch.randint(0, 1000, (300,), device="cuda", dtype=torch.int32)
    (o,) = H.gather(idx, [src])
    assert torch.equal(o, src[idx.long()])
    with pytest.raises(RuntimeError, match="n_streams"):
        H.gather(idx, [src] * 9)
    with pytest.raises(ValueError):
        H.gather(idx.long(), [src])


def test_gather_full_size_permutation_roundtrip(H):
    # BASELINE size: a full-epoch permutation gather is a bijection: scatter back == original
    B, D = 524288, 64
    src = torch.randn(B, D, device="cuda")
    rng = H.MT19937(1, B)
    perm = rng.shuffle_epochs(B, 1)[0]
    (g,) = H.gather(perm, [src])
    back = torch.empty_like(src)
    back[perm.long()] = g
    assert torch.equal(back, src)
    assert torch.equal(g.sum(dim=1).sort().values, src.sum(dim=1).sort().values)


# ---------------------------------------------------------------------------------- K4 + K5
def test_loss_golden_vs_reference_autograd(H):
    z = load("loss.npz")
    for name in z["names"]:
        T, N, norm_adv, clip_vloss, clip, ec, vc = z[f"{name}/meta"]
        a = [dev(z[f"{name}/{k}"]) for k in ("newlogp", "oldlogp", "adv", "newv", "oldv", "ret", "entropy")]
        mode = H.VLOSS_CLIPPED if clip_vloss else H.VLOSS_OLDVALUES
        sc, g_lp, g_v, g_e = H.loss_fwd_bwd(*a, clip, ec, vc, bool(norm_adv), mode)
        sc = sc.cpu().numpy()
        tol = dict(rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(sc[[1, 2, 3, 4, 5, 6]], z[f"{name}/scalars"], err_msg=name, **tol)
        np.testing.assert_allclose(g_lp.cpu().numpy(), z[f"{name}/g_newlogp"], rtol=1e-5, atol=1e-8, err_msg=name)
        np.testing.assert_allclose(g_v.cpu().numpy(), z[f"{name}/g_newv"], rtol=1e-5, atol=1e-8, err_msg=name)
        np.testing.assert_allclose(g_e.cpu().numpy(), z[f"{name}/g_entropy"], rtol=1e-6, err_msg=name)


@pytest.mark.parametrize("M", [1, 2, 63, 1000, 16384, 131072, 1048577])
@pytest.mark.parametrize("norm_adv,mode", [(True, 1), (False, 1), (True, 0), (True, 2)])
def test_loss_vs_c_oracle(H, CO, M, norm_adv, mode):
    rs = np.random.RandomState(M % 9973 + mode)
    oldlp = (-1 + 0.5 * rs.standard_normal(M)).astype(np.float32)
    newlp = (oldlp + 0.2 * rs.standard_normal(M)).astype(np.float32)
    adv = (3 * rs.standard_normal(M) + 0.5).astype(np.float32)
    oldv = rs.standard_normal(M).astype(np.float32)
    newv = (oldv + 0.3 * rs.standard_normal(M)).astype(np.float32)
    ret = (oldv + adv).astype(np.float32)
    ent = (rs.random_sample(M) + 1).astype(np.float32)
    args = (newlp, oldlp, adv, newv, oldv, ret, ent)
    sc, g_lp, g_v, g_e = H.loss_fwd_bwd(*[dev(a) for a in args], 0.2, 0.01, 0.5, norm_adv, mode)
    sc_o, glp_o, gv_o, ge_o = CO.ppo_loss(*args, 0.2, 0.01, 0.5, norm_adv, mode)
    if M == 1 and norm_adv:
        assert np.isnan(sc.cpu().numpy()[[0, 1]]).all() and np.isnan(sc_o[[0, 1]]).all()   # std of one sample
        return
    np.testing.assert_allclose(sc.cpu().numpy(), sc_o, rtol=1e-5, atol=1e-6)
    # a 1-ulp difference in expf can flip a clip-boundary sample; allow a handful at large M
    bad = ~np.isclose(g_lp.cpu().numpy(), glp_o, rtol=1e-5, atol=1e-10)
    assert bad.sum() <= max(0, M // 100000), bad.sum()
    np.testing.assert_allclose(g_v.cpu().numpy(), gv_o, rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(g_e.cpu().numpy(), ge_o, rtol=1e-6)


def test_loss_autograd_function_matches_torch_formula(H):
    M = 4096
    g = torch.Generator(device="cuda").manual_seed(0)
    oldlp = torch.randn(M, device="cuda", generator=g) - 1
    adv, oldv = torch.randn(M, device="cuda", generator=g) * 2, torch.randn(M, device="cuda", generator=g)
    ret = oldv + adv
    base = [oldlp + 0.2 * torch.randn(M, device="cuda", generator=g), oldv + 0.3 * torch.randn(M, device="cuda", generator=g),
            torch.rand(M, device="cuda", generator=g) + 1]

    def torch_loss(nl, nv, en):   # src/ppo.py:225-264 written with torch ops (fp32 reference for this kernel)
        lr = nl - oldlp
        ratio = lr.exp()
        a = (adv - adv.mean()) / (adv.std() + 1e-8)
        pg = torch.max(-a * ratio, -a * torch.clamp(ratio, 0.8, 1.2)).mean()
        vu = (nv - ret) ** 2
        vc = (oldv + torch.clamp(nv - oldv, -0.2, 0.2) - ret) ** 2
        return pg - 0.01 * en.mean() + 0.5 * (0.5 * torch.max(vu, vc).mean())
    a1 = [t.clone().requires_grad_() for t in base]
    a2 = [t.clone().requires_grad_() for t in base]
    l1 = torch_loss(*a1)
    l1.backward()
    l2 = H.ppo_loss(a2[0], a2[1].view(-1, 1), a2[2], oldlp, adv, oldv, ret, 0.2, 0.01, 0.5)
    (2.0 * l2).backward()
    torch.testing.assert_close(l2, l1, rtol=1e-5, atol=1e-6)
    for x, y in zip(a1, a2):
        torch.testing.assert_close(y.grad, 2.0 * x.grad, rtol=1e-5, atol=1e-9)


# ---------------------------------------------------------------------------------- K6
@pytest.mark.parametrize("n,scale", [(1, 1.0), (17101, 1e-3), (17101, 10.0), (2_560_000, 0.01)])
def test_grad_norm_clip_vs_torch_and_oracle(H, CO, n, scale):
    g = torch.randn(n, device="cuda") * scale
    p = torch.nn.Parameter(torch.zeros(n, device="cuda"))
    p.grad = g.clone()
    ref_norm = torch.nn.utils.clip_grad_norm_([p], 0.5)
    flat = g.clone()
    norm = H.grad_norm_clip_(flat, 0.5)
    torch.testing.assert_close(norm[0], ref_norm, rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(flat, p.grad, rtol=1e-5, atol=1e-9)
    o_flat, o_norm = CO.grad_norm_clip(g.cpu().numpy(), 0.5)
    np.testing.assert_allclose(flat.cpu().numpy(), o_flat, rtol=2e-6, atol=1e-12)


# ---------------------------------------------------------------------------------- stream semantics
def test_kernels_run_on_current_stream_and_capture_into_a_graph(H):
    T, N = 16, 64
    d = synth_rollout(T, N, 4, 2)
    r, v, dn = dev(d["rewards"]), dev(d["values"]), dev(d["terminals"])
    nv, nd = dev(d["next_value"]), dev(d["next_done"])
    ret, adv = torch.empty_like(r), torch.empty_like(r)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        H.gae(r, v, dn, nv, nd, 0.99, 0.95, out=(ret, adv))   # warm-up outside capture
    torch.cuda.current_stream().wait_stream(side)
    expect = adv.clone()
    adv.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        H.gae(r, v, dn, nv, nd, 0.99, 0.95, out=(ret, adv))
    assert float(adv.abs().sum()) == 0.0          # capture must not execute
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(adv, expect)


def test_shuffle_refuses_a_capturing_stream_and_the_generator_is_untouched(H):
    """include/aurppo.h: the shuffle family keeps host-side sequence state (slot, relay tag) and must run eagerly; a call on a
    capturing stream returns AURPPO_EINVAL instead of recording launches that would replay with a stale tag.  The refused call
    consumes nothing: the next eager shuffle is numpy's first."""
    n = 40000
    rng = H.MT19937(5, n)
    out = torch.empty((1, n), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out.zero_()                         # (something to capture: an empty graph is not the point)
        with pytest.raises(RuntimeError, match="captured"):
            rng.shuffle_epochs(n, 1, out=out)
    got = rng.shuffle_epochs(n, 1).cpu().numpy()[0]
    rs = np.random.RandomState(5)
    idx = np.arange(n)
    rs.shuffle(idx)
    np.testing.assert_array_equal(got, idx)


# ---------------------------------------------------------------------------------- packed record path
def test_gae_pack_record_and_packed_loss_match_unpacked(H, CO):
    T, N = 128, 1024
    d = synth_rollout(T, N, 4, 2)
    r, v, dn, lp = dev(d["rewards"]), dev(d["values"]), dev(d["terminals"]), dev(d["log_probs"])
    nv, nd = dev(d["next_value"]), dev(d["next_done"])
    rec = torch.empty(T * N, 4, device="cuda")
    ret, adv = H.gae(r, v, dn, nv, nd, 0.99, 0.95, log_probs=lp, rec=rec)
    ret0, adv0 = H.gae(r, v, dn, nv, nd, 0.99, 0.95)
    assert torch.equal(ret, ret0) and torch.equal(adv, adv0)
    assert torch.equal(rec, torch.stack([lp.view(-1), adv.view(-1), ret.view(-1), v.view(-1)], 1))
    # gather the record as one float4 stream and feed the packed loss: identical to the 7-stream call
    M = 4096
    idx = torch.randperm(T * N, device="cuda")[:M].int()
    (rec_mb,) = H.gather(idx, [rec])
    assert torch.equal(rec_mb, rec[idx.long()])
    newlp = rec_mb[:, 0] + 0.1 * torch.randn(M, device="cuda")
    newv = rec_mb[:, 3] + 0.2 * torch.randn(M, device="cuda")
    ent = torch.rand(M, device="cuda") + 1
    for mode in (0, 1, 2):
        a = H.loss_fwd_bwd_packed(newlp, newv, ent, rec_mb, 0.2, 0.01, 0.5, True, mode)
        b = H.loss_fwd_bwd(newlp, rec_mb[:, 0].contiguous(), rec_mb[:, 1].contiguous(), newv, rec_mb[:, 3].contiguous(),
                           rec_mb[:, 2].contiguous(), ent, 0.2, 0.01, 0.5, True, mode)
        for x, y in zip(a, b):
            assert torch.equal(x, y)


# ---------------------------------------------------------------------------------- K6b
@pytest.mark.parametrize("n,clip_n", [(17104, None), (1000, 400), (3, 3)])
def test_clip_adam_matches_torch_clip_then_adam(H, n, clip_n):
    torch.manual_seed(n)
    p0 = torch.randn(n, device="cuda")
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p_ref], lr=3e-4, eps=1e-5)
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    lr, t = torch.tensor([3e-4], device="cuda"), torch.zeros(1, device="cuda")
    for step in range(5):
        g = torch.randn(n, device="cuda") * (10.0 if step % 2 else 1e-3)
        lr.fill_(3e-4 * (1 - step / 5))
        opt.param_groups[0]["lr"] = 3e-4 * (1 - step / 5)
        p_ref.grad = g.clone()
        k = n if clip_n is None else clip_n
        # torch: clip the leading slice only, then Adam on everything
        sl = torch.nn.Parameter(torch.zeros(k, device="cuda"))
        sl.grad = p_ref.grad[:k]
        ref_norm = torch.nn.utils.clip_grad_norm_([sl], 0.5)
        opt.step()
        gg = g.clone()
        norm = H.clip_adam_(p, gg, m, v, lr, t, 0.5, clip_n)
        torch.testing.assert_close(norm[0], ref_norm, rtol=1e-5, atol=1e-9)
        torch.testing.assert_close(gg, p_ref.grad, rtol=1e-5, atol=1e-10)
        torch.testing.assert_close(p, p_ref.data, rtol=1e-6, atol=1e-7)
    assert float(t) == 5.0
    st = opt.state[p_ref]
    torch.testing.assert_close(m, st["exp_avg"], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(v, st["exp_avg_sq"], rtol=1e-5, atol=1e-12)


# ---------------------------------------------------------------------------------- K9: bias + state plane + ReLU + max-pool
@pytest.mark.parametrize("B,C,Hh,Ww,with_plane", [(3, 16, 128, 128, True), (2, 5, 84, 84, True), (4, 7, 42, 42, False),
                                                  (3, 4, 21, 21, False), (2, 3, 10, 10, False), (5, 8, 6, 6, False),
                                                  (2, 2, 7, 9, True), (1, 1, 2, 2, False)])
def test_bias_relu_pool2_matches_torch_ops_forward_and_backward(H, B, C, Hh, Ww, with_plane):
    """K9 vs the op chain it replaces (src/nets/base_cnns.py:28-45: conv bias, nn.ReLU, nn.MaxPool2d(2); plane term of
    src/models/robot_actor_critic.py:58-59) in plain PyTorch fp32, values AND gradients, odd sizes and ties included."""
    import torch.nn.functional as F
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + Hh)
    x = torch.randn(B, C, Hh, Ww, device="cuda", generator=g)
    x[0, 0].fill_(0.25)                       # a plane of ties: the first window element must win
    x[-1, -1].fill_(-1.0)                     # a dead plane
    bias = torch.randn(C, device="cuda", generator=g)
    scale = (torch.rand(B, device="cuda", generator=g) < 0.5).float() if with_plane else None
    plane = torch.randn(1, C, Hh, Ww, device="cuda", generator=g) if with_plane else None
    leaves = [t.clone().requires_grad_(True) for t in (x, bias)] + ([plane.clone().requires_grad_(True)] if with_plane else [])
    # association of base_encoder.forward_split: (conv + state * plane) + bias
    ref_in = ((leaves[0] + scale.view(-1, 1, 1, 1) * leaves[2]) if with_plane else leaves[0]) + leaves[1].view(1, -1, 1, 1)
    ref = F.max_pool2d(F.relu(ref_in), 2)
    w = torch.randn(ref.shape, device="cuda", generator=g)
    (ref * w).sum().backward()
    mine = [t.clone().requires_grad_(True) for t in (x, bias)] + ([plane.clone().requires_grad_(True)] if with_plane else [])
    y = H.bias_relu_pool2(mine[0], mine[1], scale, mine[2] if with_plane else None)
    (y * w).sum().backward()
    assert y.shape == ref.shape
    assert torch.equal(y, ref)                                  # same additions in the same order: bit-exact
    assert torch.equal(mine[0].grad, leaves[0].grad)            # routing only
    # the bias gradient sums B*Ho*Wo terms: torch in fp32 (cascade), K9 in fp64 partials -- compare at the sum's own rounding level
    n_terms = B * (Hh // 2) * (Ww // 2)
    torch.testing.assert_close(mine[1].grad, leaves[1].grad, rtol=1e-5, atol=2e-7 * n_terms ** 0.5 * float(w.abs().max()) + 1e-6)
    if with_plane:
        torch.testing.assert_close(mine[2].grad, leaves[2].grad, rtol=1e-5, atol=1e-6)


def test_fused_encoder_matches_stock_torch_encoder(H):
    """base_encoder with K9 (default on CUDA) vs the same module with fused_pool = False: outputs and every weight gradient."""
    from aur_ppo_amd.base_cnns import base_encoder, weights_init
    for shape in ((2, 128, 128), (4, 84, 84)):
        torch.manual_seed(3)
        enc = base_encoder(obs_shape=shape, out_dim=32).cuda()
        enc.apply(weights_init)
        obs = torch.rand(5, shape[0] - 1, shape[1], shape[2], device="cuda")
        state = (torch.rand(5, device="cuda") < 0.5).float()
        outs = []
        for fused, first in ((True, True), (True, False), (False, False)):
            enc.fused_pool, enc.fused_first = fused, first
            enc.zero_grad()
            f = enc.forward_split(obs, state)
            (f * torch.linspace(0.5, 1.5, f.numel(), device="cuda").view_as(f)).sum().backward()
            outs.append((f.detach().clone(), [p.grad.clone() for p in enc.parameters()]))
        for k in (0, 1):          # K10 + K9, K9 only -- each against the stock torch blocks
            torch.testing.assert_close(outs[k][0], outs[2][0], rtol=1e-4, atol=1e-5)
            for a, b in zip(outs[k][1], outs[2][1]):
                torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5 * float(b.abs().max()) + 1e-7)


# ---------------------------------------------------------------------------------- K10: the encoder's first block, fused
@pytest.mark.parametrize("B,Ci,Co,S", [(3, 1, 16, 128), (2, 3, 16, 84), (2, 1, 64, 12), (4, 2, 32, 10), (2, 1, 16, 4), (3, 3, 16, 7), (2, 2, 16, 3)])
def test_first_block_matches_conv_relu_pool_forward_and_backward(H, B, Ci, Co, S):
    """K10 vs conv2d(cat[obs, tiled state]) + ReLU + MaxPool2d(2) in plain PyTorch fp32 (src/nets/base_cnns.py:28-31 on the
    input of src/models/robot_actor_critic.py:58-59): values, weight and bias gradients; borders, odd sizes, wide blocks."""
    import torch.nn.functional as F
    g = torch.Generator(device="cuda").manual_seed(100 * B + S)
    obs = torch.rand(B, Ci, S, S, device="cuda", generator=g)
    state = (torch.rand(B, device="cuda", generator=g) < 0.5).float()
    w = (0.3 * torch.randn(Co, Ci + 1, 3, 3, device="cuda", generator=g))
    b = 0.1 * torch.randn(Co, device="cuda", generator=g)
    w1, b1 = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    x = torch.cat([obs, state.view(-1, 1, 1, 1).expand(B, 1, S, S)], 1)
    with torch.backends.cudnn.flags(enabled=False):        # torch's own direct convolution: an fp32 reference without Winograd
        ref = F.max_pool2d(F.relu(F.conv2d(x, w1, b1, padding=1)), 2)
    gy = torch.randn(ref.shape, device="cuda", generator=g)
    (ref * gy).sum().backward()
    w2, b2 = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = H.first_block(obs, state, w2, b2)
    (y * gy).sum().backward()
    assert y.shape == ref.shape
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=2e-6)
    sw, sb = float(w1.grad.abs().max()), float(b1.grad.abs().max())
    assert float((w2.grad - w1.grad).abs().max()) <= 2e-5 * sw + 1e-6, (float((w2.grad - w1.grad).abs().max()), sw)
    assert float((b2.grad - b1.grad).abs().max()) <= 2e-5 * sb + 1e-6


def test_first_block_batches_past_the_grid_limit(H):
    """gridDim.z carries the sample index (<= 65535): a larger batch goes in slices and must equal two half-batch calls."""
    B, S = 70000, 4
    g = torch.Generator(device="cuda").manual_seed(1)
    obs = torch.rand(B, 1, S, S, device="cuda", generator=g)
    state = (torch.rand(B, device="cuda", generator=g) < 0.5).float()
    w = (0.3 * torch.randn(16, 2, 3, 3, device="cuda", generator=g)).requires_grad_(True)
    b = (0.1 * torch.randn(16, device="cuda", generator=g)).requires_grad_(True)
    y = H.first_block(obs, state, w, b)
    gy = torch.randn(y.shape, device="cuda", generator=g)
    (y * gy).sum().backward()
    w2, b2 = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    h = B // 2
    ya, yb = H.first_block(obs[:h], state[:h], w2, b2), H.first_block(obs[h:], state[h:], w2, b2)
    ((ya * gy[:h]).sum() + (yb * gy[h:]).sum()).backward()
    assert torch.equal(y, torch.cat([ya, yb]))
    torch.testing.assert_close(w.grad, w2.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(b.grad, b2.grad, rtol=1e-4, atol=1e-4)
