"""GPU: summation order vs race.  K7 / K7w hand row tiles to workgroups through a counter, so the order in which a
gradient slab's sums are formed differs from launch to launch and a gradient element moves in its last bits
(finding 3).  With AURPPO_STATIC_TILES=1 the tiles are dealt by static stride instead: if the kernels'
barriers and accumulator hand-overs are sound, two launches on the same inputs must then be BIT-IDENTICAL in every word
they write (slabs, loss partials, gradients, scalars) -- also while the shuffle kernels keep the side streams busy, which is
when round 2's weight gate tripped.  Anything else would be a race.  (src/ppo.py:219-267 is what the launches compute.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B_SHUF = 524288


def _setup(layers, hidden, D, A, B, seed=0):
    from aur_ppo_amd import hip_ops as H
    from aur_ppo_amd.actor_critic import actor_critic
    from aur_ppo_amd.flat import FlatBucket
    torch.manual_seed(seed)
    pol = actor_critic(D, (A,), hidden, layers, 0.0, True).cuda()
    with torch.no_grad():
        pol.actor_logstd.copy_(0.3 * torch.randn(1, A))
        for p in pol.parameters():
            p.add_(0.05 * torch.randn_like(p))
    bucket = FlatBucket(pol.parameters())
    g = torch.Generator(device="cuda").manual_seed(seed)
    obs = torch.randn(B, D, device="cuda", generator=g)
    act = torch.randn(B, A, device="cuda", generator=g)
    with torch.no_grad():
        _, lp, _, v = pol.evaluate(obs, act)
    rec = torch.stack([lp + 0.2 * torch.randn(B, device="cuda", generator=g), 2 * torch.randn(B, device="cuda", generator=g),
                       v.view(-1) + torch.randn(B, device="cuda", generator=g),
                       v.view(-1) + 0.1 * torch.randn(B, device="cuda", generator=g)], 1).contiguous()
    rec64 = H.pack_records(rec, act)
    return H, pol, bucket, obs, rec64


def _workspace_of(H, lay):
    kind = "mlp_wide" if lay.get("wide") else "mlp"
    return H._ws_cache[(kind, torch.cuda.current_device())]


SHAPES = [(2, 64, 64, "k_mlp_step2"), (3, 64, 64, "k_mlpw_step<3,true>"), (3, 128, 64, "k_mlpw_step<3,false>"),
          (1, 64, 64, "k_mlpw_step<1,true>"), (2, 128, 128, "k_mlpw_step<2,false>")]


@pytest.mark.parametrize("layers,hidden,D,kernel", SHAPES, ids=[s[3] for s in SHAPES])
def test_static_tile_launches_are_bit_identical_with_the_side_streams_busy(layers, hidden, D, kernel, monkeypatch):
    monkeypatch.setenv("AURPPO_STATIC_TILES", "1")
    A, B, M = 6, 262144, 131072
    H, pol, bucket, obs, rec64 = _setup(layers, hidden, D, A, B)
    lay = H.mlp_layout(pol, bucket)
    assert lay is not None
    idx = torch.randperm(B, device="cuda")[:M].int()
    rng = H.MT19937(1, B_SHUF, torch.device("cuda"))
    side = torch.cuda.Stream()
    perm_out = torch.empty((4, B_SHUF), dtype=torch.int32, device="cuda")
    runs = []
    for rep in range(4):
        if rep in (1, 3):
            # the shuffle pipeline (fill / accept / link / resolve on three streams) runs beside this launch
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                rng.shuffle_epochs(B_SHUF, 4, out=perm_out)
        g = torch.full_like(bucket.flat_grad, float("nan"))
        sc = H.mlp_ppo_step(obs, None, rec64, idx, bucket.flat_param, lay, g, 0.2, 0.0, 0.5, True, 1)
        torch.cuda.synchronize()
        ws = _workspace_of(H, lay)
        runs.append((g[:lay["n_params"]].clone(), sc.clone(), ws.clone()))
    assert torch.isfinite(runs[0][0]).all() and torch.isfinite(runs[0][1]).all()
    for k in (1, 2, 3):
        assert torch.equal(runs[k][0], runs[0][0]), f"{kernel}: gradients of launch {k} differ from launch 0 in static mode"
        assert torch.equal(runs[k][1], runs[0][1]), f"{kernel}: loss scalars of launch {k} differ"
        # everything the launch left in its workspace: per-workgroup slabs, loss partials, statistics, operand copies
        assert torch.equal(runs[k][2], runs[0][2]), f"{kernel}: workspace (slabs / partials) of launch {k} differs"


@pytest.mark.parametrize("layers,hidden,D,kernel", SHAPES[:3], ids=[s[3] for s in SHAPES[:3]])
def test_counter_dealt_launches_agree_to_rounding(layers, hidden, D, kernel, monkeypatch):
    """The product mode (tiles from the counter): same inputs, side streams busy -- gradients may differ in their last
    bits only.  The bound is the measured one (tools/grad_repeatability.py): 1e-6 of the tensor's largest element."""
    monkeypatch.delenv("AURPPO_STATIC_TILES", raising=False)
    A, B, M = 6, 262144, 131072
    H, pol, bucket, obs, rec64 = _setup(layers, hidden, D, A, B)
    lay = H.mlp_layout(pol, bucket)
    idx = torch.randperm(B, device="cuda")[:M].int()
    rng = H.MT19937(1, B_SHUF, torch.device("cuda"))
    side = torch.cuda.Stream()
    perm_out = torch.empty((4, B_SHUF), dtype=torch.int32, device="cuda")
    gs = []
    for rep in range(3):
        if rep:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                rng.shuffle_epochs(B_SHUF, 4, out=perm_out)
        g = torch.full_like(bucket.flat_grad, float("nan"))
        H.mlp_ppo_step(obs, None, rec64, idx, bucket.flat_param, lay, g, 0.2, 0.0, 0.5, True, 1)
        torch.cuda.synchronize()
        gs.append(g[:lay["n_params"]].clone())
    scale = float(gs[0].abs().max())
    for k in (1, 2):
        assert float((gs[k] - gs[0]).abs().max()) <= 1e-6 * scale, (kernel, float((gs[k] - gs[0]).abs().max()), scale)
