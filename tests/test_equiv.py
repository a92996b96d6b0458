"""The build-defined C4-equivariant actor / critic (aur_ppo_amd/equiv.py; field layout of src/nets/equiv.py:12-157).
e2cnn is unavailable, so upstream numbers cannot be compared (parity unpinned); what is tested is the property that defines
these networks: rotate the observation by 90 degrees and (dx, dy) rotates by 90 degrees while p, dz, dtheta, every log-std
and the value stay put.  CPU tests here; the same property on the GPU (K9 path) in test_equiv_on_gpu."""
import numpy as np
import pytest
import torch

from aur_ppo_amd.equiv import C4Conv, EquivariantActor, EquivariantCritic, EquivariantEncoder, GroupPool


def _shift(x, r):
    """Regular-representation action of a rotation by r quarter turns on (B, fields*4, H, W): spatial rotation and the
    group channel g taking the content of g - r."""
    B, C, Hh, Ww = x.shape
    y = torch.rot90(x, r, (2, 3)).reshape(B, C // 4, 4, *([Hh, Ww] if r % 2 == 0 else [Ww, Hh]))
    return torch.roll(y, shifts=r, dims=2).reshape(B, C, *y.shape[3:])


@pytest.mark.parametrize("r", [1, 2, 3])
def test_c4conv_lifting_and_regular_are_equivariant(r):
    torch.manual_seed(0)
    lift = C4Conv(2, 3, "trivial", "regular", 3, 1)
    reg = C4Conv(3, 5, "regular", "regular", 3, 1)
    with torch.no_grad():
        lift.bias.normal_()
        reg.bias.normal_()
    x = torch.randn(2, 2, 10, 10)
    f = lift(x)
    torch.testing.assert_close(lift(torch.rot90(x, r, (2, 3))), _shift(f, r), rtol=1e-5, atol=1e-5)
    h = reg(f)
    torch.testing.assert_close(reg(_shift(f, r)), _shift(h, r), rtol=1e-5, atol=1e-5)
    inv = GroupPool()(h)
    torch.testing.assert_close(GroupPool()(_shift(h, r)), torch.rot90(inv, r, (2, 3)), rtol=1e-5, atol=1e-5)


def _rot_vec(v, r):
    for _ in range(r):
        v = torch.stack([-v[:, 1], v[:, 0]], 1)
    return v


@pytest.mark.parametrize("size,ch", [(128, 2), (84, 4)])
def test_actor_and_critic_equivariance(size, ch):
    torch.manual_seed(1)
    actor = EquivariantActor(obs_shape=(ch, size, size), action_dim=5, n_hidden=8)
    critic = EquivariantCritic(obs_shape=(ch, size, size), n_hidden=8)
    with torch.no_grad():
        for m in list(actor.modules()) + list(critic.modules()):
            if isinstance(m, C4Conv):
                m.bias.normal_(0, 0.1)
        actor.b_inv.normal_(0, 0.1)
    obs = torch.rand(3, ch, size, size)
    obs[:, -1] = (torch.rand(3) < 0.5).float().view(3, 1, 1)             # the tiled gripper state: a constant plane
    with torch.no_grad():
        mean, log_std = actor(obs)
        v = critic(obs)
        assert mean.shape == (3, 5) and log_std.shape == (3, 5) and v.shape == (3, 1, 1, 1)
        assert torch.equal(v.tensor, v) and type(v.tensor) is torch.Tensor     # the GeometricTensor-style accessor
        assert float(mean[:, 1:3].abs().max()) > 1e-4
        for r in (1, 2, 3):
            m2, ls2 = actor(torch.rot90(obs, r, (2, 3)))
            torch.testing.assert_close(m2[:, 1:3], _rot_vec(mean[:, 1:3], r), rtol=1e-4, atol=1e-5)      # (dx, dy) turns
            torch.testing.assert_close(m2[:, [0, 3, 4]], mean[:, [0, 3, 4]], rtol=1e-4, atol=1e-5)       # p, dz, dtheta do not
            torch.testing.assert_close(ls2, log_std, rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(critic(torch.rot90(obs, r, (2, 3))).tensor, v.tensor, rtol=1e-4, atol=1e-5)


def test_encoder_field_layout_matches_reference_widths():
    enc = EquivariantEncoder(2, 128, 128)                       # src/nets/equiv.py:17-58: 16, 32, 64, 128, 256, 128, 128 fields
    assert [b.conv.out_fields for b in enc.conv] == [16, 32, 64, 128, 256, 128, 128]
    assert [b.pool for b in enc.conv] == [2, 2, 2, 2, 0, 2, 0]
    assert [b.conv.padding for b in enc.conv] == [1, 1, 1, 1, 1, 0, 0]
    with pytest.raises(ValueError):
        EquivariantEncoder(2, 128, 100)


def test_robot_actor_critic_equivariant_branch_and_trainer_on_cpu():
    from aur_ppo_amd.robot_actor_critic import robot_actor_critic
    from aur_ppo_amd.robot_ppo import robot_ppo
    from tests import oracle_ops
    from tests.test_robot_host import _params
    torch.manual_seed(2)
    pol = robot_actor_critic(torch.device("cpu"), True, n_hidden=8)
    assert not hasattr(pol, "actor_logstd")                      # src/models/robot_actor_critic.py:33-35
    s, o = (torch.rand(2) < 0.5).float(), torch.rand(2, 1, 128, 128)
    acts, unscaled, lp, ent, v = pol.evaluate(s, o)
    assert acts.shape == (2, 5) and lp.shape == (2,) and ent.shape == (2,) and v.tensor.shape == (2, 1, 1, 1)
    assert pol.value(s, o).flatten().shape == (2,)
    a = robot_ppo(_params(equivariant=True, equiv_hidden=8, num_steps=4, total_timesteps=16), ops=oracle_ops)
    before = a.bucket.flat_param.clone()
    a.train()
    assert np.isfinite(a._last_scalars).all() and not torch.equal(before, a.bucket.flat_param)


@pytest.mark.gpu
def test_equiv_on_gpu():
    """Same property through the GPU path (MIOpen convolutions + K9 blocks), and one robot_ppo update with the
    equivariant policy on the HIP kernels."""
    torch.manual_seed(3)
    actor = EquivariantActor(obs_shape=(2, 128, 128), action_dim=5, n_hidden=32).cuda()      # first block: 4 fields x 4 = 16 channels (K10)
    critic = EquivariantCritic(obs_shape=(2, 128, 128), n_hidden=32).cuda()
    obs = torch.rand(4, 2, 128, 128, device="cuda")
    obs[:, 1] = (torch.rand(4, device="cuda") < 0.5).float().view(4, 1, 1)       # channel 1: the tiled gripper state
    with torch.no_grad():
        mean, log_std = actor(obs)
        v = critic(obs)
        m2, ls2 = actor(torch.rot90(obs, 1, (2, 3)))
        torch.testing.assert_close(m2[:, 1:3], _rot_vec(mean[:, 1:3], 1), rtol=1e-3, atol=1e-5)
        torch.testing.assert_close(m2[:, [0, 3, 4]], mean[:, [0, 3, 4]], rtol=1e-3, atol=1e-5)
        torch.testing.assert_close(ls2, log_std, rtol=1e-3, atol=1e-5)
        torch.testing.assert_close(critic(torch.rot90(obs, 1, (2, 3))).tensor, v.tensor, rtol=1e-3, atol=1e-5)
        # K10 first block (image and state given separately) == K9 blocks on the concatenated input == stock torch blocks
        m4, ls4 = actor(obs[:, :1].contiguous(), obs[:, 1, 0, 0].contiguous())
        torch.testing.assert_close(m4, mean, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(critic(obs[:, :1].contiguous(), obs[:, 1, 0, 0].contiguous()).tensor, v.tensor, rtol=1e-4, atol=1e-6)
        for b in actor.enc.conv:
            b.fused_pool = False
        m3, _ = actor(obs)
        torch.testing.assert_close(m3, mean, rtol=1e-4, atol=1e-6)
    from aur_ppo_amd.robot_ppo import robot_ppo
    from aur_ppo_amd.robot_run import build_parser, params_from_args
    p = params_from_args(build_parser().parse_args([]))
    p.update(gym_id="Synthetic-arm", num_envs=4, num_steps=4, total_timesteps=32, num_update_epochs=2, num_minibatches=2,
             do_pretraining=False, log=False, equivariant=True, equiv_hidden=32)
    a = robot_ppo(p)
    a.train()
    assert np.isfinite(a._last_scalars).all() and a._last_scalars.shape == (4, 9)
