"""GPU: BASELINE configs 3 and 5 at THEIR workload through the kernels that do not need a convolution library.

config 3: close_loop_block_picking, (1,128,128) observations, num_envs 256, T 128  -> B 32 768, minibatches of 8 192
config 5: 84x84x3 observations, num_envs 2048 over 8 GPUs = 256 per GPU, T 64      -> B 16 384, minibatches of 4 096
(src/robot_ppo.py:224-244 GAE that never visits t = T-1, :329-408 update: shuffle, slicing gathers, first encoder block of
src/nets/base_cnns.py:20-54).  The whole robot update at these sizes spends minutes in MIOpen's first-use search on a fresh
box, so the pieces are exercised one by one here at full size, and the whole update at the configs' env count
in tests/test_robot_gpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CONFIGS = [pytest.param(128, 256, 1, 128, 8192, id="config3_T128_N256_1x128x128"),
           pytest.param(64, 256, 3, 84, 4096, id="config5_shard_T64_N256_3x84x84")]


@pytest.mark.parametrize("T,N,C,S,M", CONFIGS)
def test_k1_skip_last_bit_exact_at_config_shape(T, N, C, S, M):
    from aur_ppo_amd import hip_ops as H
    from oracle import ppo_oracle as O
    rs = np.random.RandomState(T + N)
    r = (rs.random_sample((T, N)) < 0.3).astype(np.float32)            # sparse 0/1 rewards, as the arm tasks give
    v = rs.standard_normal((T, N)).astype(np.float32)
    d = (rs.random_sample((T, N)) < 0.05).astype(np.float32)
    nv, nd = rs.standard_normal(N).astype(np.float32), (rs.random_sample(N) < 0.05).astype(np.float32)
    g = lambda a: torch.from_numpy(a).cuda()
    ret, adv = H.gae(g(r), g(v), g(d), g(nv), g(nd), 0.99, 0.95, H.GAE_SKIP_LAST)
    ret_o, adv_o = O.gae(r, v, d, nv, nd, 0.99, 0.95, O.GAE_MODE_SKIP_LAST)
    assert np.array_equal(adv.cpu().numpy(), adv_o) and np.array_equal(ret.cpu().numpy(), ret_o)
    assert float(adv[-1].abs().max()) == 0.0            # upstream's loop never visits the last step


@pytest.mark.parametrize("T,N,C,S,M", CONFIGS)
def test_k2_epoch_permutations_bit_exact_at_config_batch(T, N, C, S, M):
    from aur_ppo_amd import hip_ops as H
    from oracle import ppo_oracle as O
    B, E = T * N, 4
    rng = H.MT19937(1, B, torch.device("cuda"))
    got = rng.shuffle_epochs(B, E).cpu().numpy()
    ref = O.epoch_permutations(1, B, E)
    for e in range(E):
        assert np.array_equal(got[e], ref[e]), f"epoch {e}"
    key, pos = rng.get_state()
    st = np.random.RandomState(1)
    idx = np.arange(B)
    for _ in range(E):
        st.shuffle(idx)
    k2 = st.get_state()
    assert np.array_equal(np.asarray(key, dtype=np.uint32), k2[1]) and int(pos) == int(k2[2])


@pytest.mark.parametrize("T,N,C,S,M", CONFIGS)
def test_k3_gathers_image_rows_bit_exact_at_config_minibatch(T, N, C, S, M):
    """The minibatch slicing gathers of src/robot_ppo.py:341-353 over the whole rollout buffer of the config: image rows of
    C*S*S floats (64 KB / 83 KB each), the gripper state, the 5-float action rows and the packed record, M rows at once."""
    from aur_ppo_amd import hip_ops as H
    B = T * N
    g = torch.Generator(device="cuda").manual_seed(B)
    obs = torch.rand(B, C, S, S, device="cuda", generator=g)
    state = (torch.rand(B, device="cuda", generator=g) < 0.5).float()
    act = torch.randn(B, 5, device="cuda", generator=g)
    rec = torch.randn(B, 4, device="cuda", generator=g)
    idx = torch.randperm(B, device="cuda", generator=g)[:M].int()
    outs = H.gather(idx, [obs, state, act, rec])
    li = idx.long()
    for o, s in zip(outs, (obs, state, act, rec)):
        assert o.shape[0] == M and torch.equal(o, s[li])


@pytest.mark.parametrize("T,N,C,S,M", CONFIGS)
def test_first_block_at_config_minibatch_matches_torch_on_slices(T, N, C, S, M):
    """K10 (conv 3x3 over [image, tiled state] + ReLU + 2x2 max-pool, forward and weight / bias gradients) on a whole
    minibatch of the config (the full-batch torch reference would materialise the 8.6 GB pre-pool tensor this kernel
    exists to avoid, so:)
      * values against torch's direct fp32 convolution on 8 chunks of 64 samples taken across the batch;
      * gradients: the upstream gradient is zero outside those chunks, so the full-batch result must equal the SUM of the
        same kernel's results on the chunks alone (a per-sample kernel: same decisions, fixed-order sums) to rounding;
      * each chunk's gradient against torch's on that chunk, with the upstream gradient zeroed on the (handful of) pooling
        windows whose two largest pre-pool candidates -- or whose winner and the ReLU threshold -- are within 1e-5 of each
        other: there two correct fp32 implementations may route the gradient to different positions (each such window
        moves a weight-gradient element by up to |gy * x| ~ 4, against 2e-5 x 600 for rounding), everywhere else they must
        agree to the tolerance the small shapes of tests/test_hip_parity.py hold."""
    import torch.nn.functional as F
    from aur_ppo_amd import hip_ops as H
    g = torch.Generator(device="cuda").manual_seed(M)
    obs = torch.rand(M, C, S, S, device="cuda", generator=g)
    state = (torch.rand(M, device="cuda", generator=g) < 0.5).float()
    w = (0.3 * torch.randn(16, C + 1, 3, 3, device="cuda", generator=g)).requires_grad_(True)
    b = (0.1 * torch.randn(16, device="cuda", generator=g)).requires_grad_(True)
    y = H.first_block(obs, state, w, b)
    assert y.shape == (M, 16, S // 2, S // 2)
    chunks = [(k * (M // 8), k * (M // 8) + 64) for k in range(8)]
    gy = torch.zeros_like(y)
    for lo, hi in chunks:
        gy[lo:hi] = torch.randn(hi - lo, 16, S // 2, S // 2, device="cuda", generator=g)
    (y * gy).sum().backward()
    sum_w, sum_b = torch.zeros_like(w), torch.zeros_like(b)
    with torch.backends.cudnn.flags(enabled=False):
        for lo, hi in chunks:
            wk, bk = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
            yk = H.first_block(obs[lo:hi].contiguous(), state[lo:hi].contiguous(), wk, bk)
            assert torch.equal(yk, y[lo:hi])
            (yk * gy[lo:hi]).sum().backward()
            sum_w += wk.grad
            sum_b += bk.grad
            w1, b1 = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
            x = torch.cat([obs[lo:hi], state[lo:hi].view(-1, 1, 1, 1).expand(hi - lo, 1, S, S)], 1)
            z = F.conv2d(x, w1, b1, padding=1)
            ref = F.max_pool2d(F.relu(z), 2)
            torch.testing.assert_close(y[lo:hi], ref, rtol=1e-5, atol=2e-6)
            with torch.no_grad():        # windows without a clear winner
                So = S // 2
                win = z[:, :, :2 * So, :2 * So].reshape(hi - lo, 16, So, 2, So, 2).permute(0, 1, 2, 4, 3, 5).reshape(hi - lo, 16, So, So, 4)
                top = win.topk(2, dim=-1).values
                clear = ((top[..., 0] - top[..., 1]) > 1e-5 * top[..., 0].abs().clamp(min=1.0)) & (top[..., 0].abs() > 1e-5)
                gm = gy[lo:hi] * clear
            (ref * gm).sum().backward()
            wk2, bk2 = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
            (H.first_block(obs[lo:hi].contiguous(), state[lo:hi].contiguous(), wk2, bk2) * gm).sum().backward()
            assert float((~clear).float().mean()) < 1e-4                      # a handful per chunk, not a loophole
            sw, sb = float(w1.grad.abs().max()), float(b1.grad.abs().max())
            assert float((wk2.grad - w1.grad).abs().max()) <= 2e-5 * sw + 1e-6, (float((wk2.grad - w1.grad).abs().max()), sw)
            assert float((bk2.grad - b1.grad).abs().max()) <= 2e-5 * sb + 1e-6, (float((bk2.grad - b1.grad).abs().max()), sb)
    sw, sb = float(sum_w.abs().max()), float(sum_b.abs().max())
    assert float((w.grad - sum_w).abs().max()) <= 2e-5 * sw + 1e-6, (float((w.grad - sum_w).abs().max()), sw)
    assert float((b.grad - sum_b).abs().max()) <= 2e-5 * sb + 1e-6, (float((b.grad - sum_b).abs().max()), sb)


@pytest.mark.parametrize("T,N,C,S,M", CONFIGS)
def test_bias_relu_pool_at_config_minibatch_matches_oracle_on_slices(T, N, C, S, M):
    """K9 on the second block's input shape of the config's minibatch ((M, 32, S/2, S/2)): forward bit-exact against the
    numpy checker on slices, input gradient = routing of dy through the arg-max mask (checked against torch on the slices)."""
    import torch.nn.functional as F
    from aur_ppo_amd import hip_ops as H
    from oracle import ppo_oracle as O
    Hs = S // 2
    g = torch.Generator(device="cuda").manual_seed(M + 1)
    x = torch.randn(M, 32, Hs, Hs, device="cuda", generator=g).requires_grad_(True)
    bias = torch.randn(32, device="cuda", generator=g)
    y = H.bias_relu_pool2(x, bias)
    gy = torch.randn(y.shape, device="cuda", generator=g)
    (y * gy).sum().backward()
    for lo in (0, M // 2 - 3, M - 8):
        xs = x.detach()[lo:lo + 8]
        ref = O.bias_relu_pool2(xs.cpu().numpy(), bias.cpu().numpy())
        assert np.array_equal(y[lo:lo + 8].detach().cpu().numpy(), ref)
        xt = xs.clone().requires_grad_(True)
        (F.max_pool2d(F.relu(xt + bias.view(1, -1, 1, 1)), 2) * gy[lo:lo + 8]).sum().backward()
        assert torch.equal(x.grad[lo:lo + 8], xt.grad)
