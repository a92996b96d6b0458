"""GPU: K11 (csrc/conv.hip) -- the hidden 3x3 convolutions of the robot policy's encoder (src/nets/base_cnns.py:32-45) as an implicit
GEMM on bf16 MFMAs over three-way splits -- against a plain PyTorch fp32 reference of the same op (torch's conv2d with MIOpen's
Winograd solvers disabled would still be a library kernel: the reference here is the fp64 convolution, which any fp32
implementation meets to a few ulps (2^-24 = 6e-8) of the sum of |products| -- the bound is 1e-6; tools/k11_fuzz.py measures
3.7e-7 at worst over random shapes and prints the library's own fp32 convolution on the same metric beside it)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, w, pad):
    return torch.nn.functional.conv2d(x.double(), w.double(), None, stride=1, padding=pad)


# (B, Ci, Co, H, W, pad): the encoder's hidden layers at a small batch (config 3 and config 5 sizes), odd sizes, a batch that does
# not fill a 32-pixel block, output channels that are not a multiple of 32, pixel blocks that straddle image boundaries
SHAPES = [(4, 16, 32, 64, 64, 1), (4, 32, 64, 32, 32, 1), (4, 64, 128, 16, 16, 1), (4, 128, 256, 8, 8, 1), (4, 256, 256, 8, 8, 0),
          (8, 256, 128, 3, 3, 0), (3, 16, 32, 42, 42, 1), (3, 32, 64, 21, 21, 1), (3, 64, 128, 10, 10, 1), (5, 128, 256, 5, 5, 0),
          (1, 16, 16, 5, 7, 1), (2, 32, 48, 9, 4, 0), (1, 48, 80, 6, 6, 2), (7, 16, 96, 3, 3, 1)]


@pytest.mark.parametrize("B,Ci,Co,H,W,pad", SHAPES)
def test_conv3x3_forward_and_gradients_match_the_fp64_convolution(B, Ci, Co, H, W, pad):
    from aur_ppo_amd import hip_ops as Hh
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + Ci + Co + H)
    x = torch.randn(B, Ci, H, W, device="cuda", generator=g, requires_grad=True)
    w = (torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) * (2.0 / (9 * Ci)) ** 0.5).requires_grad_(True)
    z = Hh.conv3x3(x, w, pad)
    zr = _ref(x.detach(), w.detach(), pad)
    assert z.shape == zr.shape
    # scale of the rounding: the sum of |products| behind an output element
    mag = torch.nn.functional.conv2d(x.detach().abs().double(), w.detach().abs().double(), None, padding=pad)
    err = ((z.double() - zr).abs() / mag.clamp_min(1e-30)).max().item()
    assert err <= 1e-6, f"forward: {err:.3e} of sum|ab|"
    gz = torch.randn(z.shape, device="cuda", generator=g)
    z.backward(gz)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    torch.nn.functional.conv2d(xd, wd, None, padding=pad).backward(gz.double())
    # input gradient: K11 (mode 1); scale: sum |g| |w| behind an input element
    magx = torch.nn.functional.conv_transpose2d(gz.abs().double(), w.detach().abs().double(), None, padding=pad)
    errx = ((x.grad.double() - xd.grad).abs() / magx.clamp_min(1e-30)).max().item()
    assert errx <= 1e-6, f"input gradient: {errx:.3e} of sum|ab|"
    # weight gradient: the library's kernel behind the same autograd node (fp32 sums over B * H * W terms)
    np.testing.assert_allclose(w.grad.cpu().numpy(), wd.grad.float().cpu().numpy(), rtol=2e-4, atol=2e-5 * float(wd.grad.abs().max()))


# K12: batches large enough for several slices of pixels, slices that begin inside an image, maps whose size is not a multiple of
# 4 (pixel quads that straddle rows and images), channel counts that do not fill the 128 x 128 tile, the 64-channel tile shape
WGRAD_SHAPES = SHAPES + [(64, 64, 128, 16, 16, 1), (48, 32, 64, 21, 21, 1), (33, 128, 256, 5, 5, 0), (40, 64, 64, 10, 10, 1),
                         (16, 32, 160, 13, 11, 2), (256, 256, 256, 8, 8, 0)]


@pytest.mark.parametrize("B,Ci,Co,H,W,pad", WGRAD_SHAPES)
def test_conv3x3_weight_gradient_matches_the_fp64_convolution(B, Ci, Co, H, W, pad, monkeypatch):
    """K12 (csrc/conv.hip::k_conv3x3_wgrad) against the fp64 weight gradient, at the fp32 rounding level of the sum of |products|
    behind a filter element; and through the autograd node of ``conv3x3`` (AURPPO_K12_ALL=1: every shape takes K12)."""
    from aur_ppo_amd import hip_ops as Hh
    g = torch.Generator(device="cuda").manual_seed(B * 977 + Ci + 3 * Co + H)
    x = torch.randn(B, Ci, H, W, device="cuda", generator=g)
    gz = torch.randn(B, Co, H + 2 * pad - 2, W + 2 * pad - 2, device="cuda", generator=g)
    dw = Hh.conv3x3_wgrad(gz, x, Co, pad)
    assert dw.shape == (Co, Ci, 3, 3)

    def wgrad64(xx, gg):
        wd = torch.zeros(Co, Ci, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True)
        torch.nn.functional.conv2d(xx.double(), wd, None, padding=pad).backward(gg.double())
        return wd.grad
    ref, mag = wgrad64(x, gz), wgrad64(x.abs(), gz.abs())
    err = ((dw.double() - ref).abs() / mag.clamp_min(1e-30)).max().item()
    assert err <= 1e-6, f"weight gradient: {err:.3e} of sum|ab|"
    # the same through autograd, next to the forward pass and the input gradient
    monkeypatch.setenv("AURPPO_K12_ALL", "1")
    if Ci % 16 == 0:
        xin = x.clone().requires_grad_(True)
        w = (torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) * (2.0 / (9 * Ci)) ** 0.5).requires_grad_(True)
        Hh.conv3x3(xin, w, pad).backward(gz)
        assert torch.equal(w.grad, dw)
    # determinism: the slices are summed in slice order
    assert torch.equal(Hh.conv3x3_wgrad(gz, x, Co, pad), dw)


@pytest.mark.parametrize("M,N,K", [(32768, 256, 256), (40001, 256, 64), (33000, 96, 256), (70001, 160, 48), (32768, 64, 64), (1000, 256, 256)])
def test_linear_weight_gradient_matches_the_fp64_product(M, N, K):
    """csrc/conv.hip::k_linear_wgrad (dW = dY^T X for the layers wider than the fused steps cover) against fp64."""
    from aur_ppo_amd import hip_ops as Hh
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    gy = torch.randn(M, N, device="cuda", generator=g)
    x = torch.randn(M, K, device="cuda", generator=g)
    dw = Hh.linear_wgrad(gy, x)
    ref = gy.double().t() @ x.double()
    mag = gy.abs().double().t() @ x.abs().double()
    err = ((dw.double() - ref).abs() / mag).max().item()
    assert err <= 1e-6, f"{err:.3e} of sum|ab|"
    assert torch.equal(Hh.linear_wgrad(gy, x), dw)


def test_conv3x3_equals_torch_conv2d_to_2e5_on_encoder_data():
    """The tolerance K10 is held to (2e-5 against torch's own fp32 convolution), at a hidden block's shape."""
    from aur_ppo_amd import hip_ops as Hh
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.rand(16, 32, 32, 32, device="cuda", generator=g)
    w = torch.randn(64, 32, 3, 3, device="cuda", generator=g) * 0.08
    z = Hh.conv3x3(x, w, 1)
    zt = torch.nn.functional.conv2d(x, w, None, padding=1)
    torch.testing.assert_close(z, zt, rtol=2e-5, atol=2e-5)


def test_encoder_with_k11_matches_the_stock_encoder(monkeypatch):
    """base_encoder with K11 for its hidden convolutions against the same module on torch's convolutions: features and every
    parameter gradient."""
    from aur_ppo_amd import hip_ops as Hh
    from aur_ppo_amd.base_cnns import base_encoder
    monkeypatch.setattr(Hh, "CONV3X3_MIN_PIXELS", 1)       # (the size rule would hand this small batch to the library)
    monkeypatch.setattr(Hh, "K12_MIN_PIXELS", 1)           # ... and its weight gradients (K12)
    torch.manual_seed(0)
    enc = base_encoder(obs_shape=(2, 128, 128), out_dim=128).cuda()
    obs = torch.rand(6, 1, 128, 128, device="cuda")
    state = (torch.rand(6, device="cuda") < 0.5).float()
    outs = []
    for fused in (True, False):
        enc.fused_conv = fused
        enc.zero_grad()
        y = enc.forward_split(obs, state)
        y.square().sum().backward()
        outs.append((y.detach().clone(), [p.grad.detach().clone() for p in enc.parameters()]))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=2e-5, atol=2e-5)
    for a, b in zip(outs[0][1], outs[1][1]):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5 * float(b.abs().max()) + 1e-7)


@pytest.mark.parametrize("M,K,N", [(32768, 64, 256), (40000, 256, 256), (33000, 256, 96), (32768, 16, 32), (70001, 48, 160)])
def test_linear_forward_and_input_gradient_match_the_fp64_product(M, K, N, monkeypatch):
    """csrc/conv.hip::k_linear (nn.Linear's product for the MLP shapes the fused steps do not cover) against the fp64 matrix product:
    the fp32 rounding level of the sum of |products|; through nets._Linear the module's forward / backward use it."""
    from aur_ppo_amd import hip_ops as Hh
    from aur_ppo_amd import nets
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * (1.0 / K) ** 0.5
    y = Hh.linear_nobias(x, w, 0)
    yr = x.double() @ w.double().t()
    mag = x.abs().double() @ w.abs().double().t()
    assert ((y.double() - yr).abs() / mag).max().item() <= 1e-6
    if N % 16 == 0 and K % 32 == 0:
        gy = torch.randn(M, N, device="cuda", generator=g)
        gx = Hh.linear_nobias(gy, w, 1)
        gxr = gy.double() @ w.double()
        magx = gy.abs().double() @ w.abs().double()
        assert ((gx.double() - gxr).abs() / magx).max().item() <= 1e-6
    # the module: same parameters, K-linear forward + input gradient, split-batch weight gradient
    monkeypatch.setenv("AURPPO_LINEAR_BF16X3", "1")
    lin = nets._Linear(K, N).cuda()
    xin = x.clone().requires_grad_(True)
    out = lin(xin)
    out.square().sum().backward()
    ref = torch.nn.Linear(K, N).cuda()
    ref.load_state_dict(lin.state_dict())
    xr = x.clone().requires_grad_(True)
    outr = ref(xr)
    outr.square().sum().backward()
    torch.testing.assert_close(out, outr, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(xin.grad, xr.grad, rtol=2e-4, atol=2e-5 * float(xr.grad.abs().max()))
    torch.testing.assert_close(lin.weight.grad, ref.weight.grad, rtol=2e-4, atol=2e-5 * float(ref.weight.grad.abs().max()))


def test_fused_linear_tanh_layer_matches_the_stock_modules(monkeypatch):
    """nets._TanhMLP with AURPPO_LINEAR_BF16X3=1: (Linear, Tanh) pairs as one kernel -- outputs and every gradient against the same
    Sequential on torch's modules (same parameters)."""
    from aur_ppo_amd import nets
    torch.manual_seed(3)
    net = nets.continuous_net(256, (64,), (6,), 2, 0.0).cuda()
    x = torch.randn(32768, 64, device="cuda")
    outs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("AURPPO_LINEAR_BF16X3", fused)
        net.zero_grad()
        xin = x.clone().requires_grad_(True)
        y = net(xin)
        (y.square().sum() / y.numel()).backward()
        outs.append((y.detach().clone(), xin.grad.clone(), [p.grad.clone() for p in net.parameters()]))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=2e-4, atol=2e-5 * float(outs[1][1].abs().max()))
    for a, b in zip(outs[0][2], outs[1][2]):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5 * float(b.abs().max()) + 1e-9)
