"""GPU: robot_ppo (image observations, CNN policy) on the HIP path vs the CPU restatement."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _update_vs_cpu_restatement(C, S):
    """One ``robot_ppo.update`` (2 epochs x 2 minibatches = 4 optimizer steps) on the HIP path and the same update through
    ``oracle.reference_robot_update`` (src/robot_ppo.py:329-408) from the same weights, data and shuffle seed.  Returns the
    per-step scalar rows of both, update()'s 6-tuple, the hyper-parameters and both final state dicts."""
    from aur_ppo_amd.robot_actor_critic import robot_actor_critic
    from aur_ppo_amd.robot_ppo import robot_ppo
    from aur_ppo_amd.robot_run import build_parser, params_from_args
    from oracle import ppo_oracle as O
    p = params_from_args(build_parser().parse_args([]))
    p.update(gym_id="Synthetic-arm", num_envs=8, num_steps=8, total_timesteps=128, num_update_epochs=2, num_minibatches=2,
             do_pretraining=False, log=False, clip_vloss=True, entropy_coeff=0.01, obs_size=S, obs_channels=C)
    torch.manual_seed(2)
    agent = robot_ppo(p)
    assert agent.device.type == "cuda"
    cpu = robot_actor_critic(torch.device("cpu"), False, obs_shape=(C, S, S))
    cpu.load_state_dict({k: v.cpu() for k, v in agent.policy.state_dict().items()})
    g = torch.Generator().manual_seed(9)
    T, N = 8, 8
    buf = dict(states=(torch.rand(T, N, generator=g) < 0.5).float(), observations=torch.rand(T, N, C, S, S, generator=g),
               actions=0.3 * torch.randn(T, N, 5, generator=g), true_actions=torch.zeros(T, N, 5),
               rewards=(torch.rand(T, N, generator=g) < 0.3).float(), terminals=(torch.rand(T, N, generator=g) < 0.1).float())
    with torch.no_grad():
        _, _, lp, _, v = cpu.evaluate(buf["states"].view(-1), buf["observations"].view(-1, C, S, S), buf["actions"].view(-1, 5))
    buf["log_probs"] = (lp.view(T, N) + 0.05 * torch.randn(T, N, generator=g))
    buf["values"] = v.view(T, N).clone()
    for k, t in buf.items():
        getattr(agent.buffer, k).copy_(t)
    next_state, next_obs = (torch.rand(N, generator=g) < 0.5).float(), torch.rand(N, C, S, S, generator=g)
    next_done = torch.zeros(N)
    agent.seed_all(1)
    ret, adv = agent.advantages(next_state.cuda(), next_obs.cuda(), next_done.cuda(), agent.buffer, T)
    # skip-last GAE as upstream (F4): checked against the oracle given the CPU bootstrap value
    with torch.no_grad():
        nv = cpu.value(next_state, next_obs).flatten()
    ret_o, adv_o = O.gae(buf["rewards"].numpy(), buf["values"].numpy(), buf["terminals"].numpy(), nv.numpy(),
                         next_done.numpy(), 0.99, 0.95, O.GAE_MODE_SKIP_LAST)
    np.testing.assert_allclose(adv.cpu().numpy(), adv_o, atol=1e-5)
    assert float(adv[-1].abs().max()) == 0.0
    out = agent.update(agent.buffer.flatten(ret, adv), 2, agent.batch_size, agent.minibatch_size, [])
    flat_cpu = (buf["states"].view(-1), buf["observations"].view(-1, C, S, S), buf["log_probs"].reshape(-1),
                buf["actions"].view(-1, 5), torch.from_numpy(adv_o).reshape(-1), torch.from_numpy(ret_o).reshape(-1),
                buf["values"].reshape(-1), buf["true_actions"].view(-1, 5))
    opt = torch.optim.Adam(cpu.parameters(), lr=p["learning_rate"], eps=1e-5)
    rows = O.reference_robot_update(cpu, opt, flat_cpu, p, np.random.RandomState(1), agent.minibatch_size)
    got = agent._last_scalars
    assert got.shape[0] == rows.shape[0] == 4
    return (got.copy(), rows.copy(), [float(x) for x in out[:3]], dict(p),
            {k: v.detach().cpu().numpy() for k, v in agent.policy.state_dict().items()},
            {k: v.detach().numpy() for k, v in cpu.state_dict().items()})


# BASELINE.json's north star: "losses within 1e-5".  Optimizer step 1 of the update runs on IDENTICAL weights on both sides, so
# its six loss scalars (loss, policy loss, value loss, entropy, old_approx_kl, approx_kl) are a pure statement about the forward
# pass, the GAE and the loss arithmetic: held to rtol 1e-5 (+ 1e-6 absolute for the two KL estimates, which are differences of
# O(1) numbers that come out at 1e-3) -- ten times tighter in absolute terms than the north star asks.
STEP1_RTOL, STEP1_ATOL = 1e-5, 1e-6
# From step 2 on the weights differ: Adam's update is lr * m / (sqrt(v) + eps) with eps = 1e-5, so a gradient element far below
# eps moves its weight by lr * g / eps -- a sensitivity of lr / eps = 30 to absolute gradient differences -- and an element at
# rounding-noise level (|g| ~ 1e-7 relative to the tensor's largest, the order in which CPU direct convolution and MIOpen's
# kernels form their sums) gets a step of up to +-lr with EITHER sign.  After k steps two correct implementations can sit
# k * lr = 3e-4 .. 1.2e-3 apart in such weights; with ~1e5 conv weights feeding O(1) activations that is a relative 1e-4 in the
# heads' outputs, which is the 2e-4 the later steps' scalars are held to (tests/test_parity_fullsize.py derives the same bound
# for the MLP policy, where the gradient noise is 100x smaller and 1e-5 holds throughout).
LATER_RTOL, LATER_ATOL = 2e-4, 2e-5


def _check_later_steps_and_weights(got, rows, out, p, sd_gpu, sd_cpu):
    np.testing.assert_allclose(got[1:, :6], rows[1:, :6], rtol=LATER_RTOL, atol=LATER_ATOL)
    np.testing.assert_allclose(out[1], rows[-1, 2] * p["value_coeff"], rtol=LATER_RTOL, atol=2e-6)
    for k in sd_gpu:
        d = np.abs(sd_gpu[k] - sd_cpu[k])
        assert d.max() <= 1.2e-3, (k, d.max())      # 4 steps x lr 3e-4
        if d.size >= 1000:
            assert np.mean(d > 3e-5) < 0.06, (k, np.mean(d > 3e-5))


@pytest.mark.parametrize("C,S", [(1, 128), (3, 84)], ids=["128x128x1_reference", "84x84x3_build_defined"])
def test_robot_update_matches_cpu_restatement(C, S):
    """Product configuration (MIOpen picks its own solvers -- Winograd for the 3x3 blocks).  Step 1: 1e-5; later steps: 2e-4 with the
    derivation above."""
    got, rows, out, p, sd_gpu, sd_cpu = _update_vs_cpu_restatement(C, S)
    np.testing.assert_allclose(got[0, :6], rows[0, :6], rtol=STEP1_RTOL, atol=STEP1_ATOL)
    _check_later_steps_and_weights(got, rows, out, p, sd_gpu, sd_cpu)


def test_robot_update_with_the_hand_written_convolutions_matches_cpu_restatement(monkeypatch):
    """K11 (csrc/conv.hip: the hidden 3x3 convolutions on bf16 MFMAs over three-way splits, forward and input gradient) takes
    over from the library only above ~512 workgroups of pixels; this update is far smaller, so the size rule is lifted here: the
    same comparison with K11 carrying every hidden convolution it supports -- step 1 at 1e-5, later steps at 2e-4."""
    from aur_ppo_amd import hip_ops as Hh
    monkeypatch.setattr(Hh, "CONV3X3_MIN_PIXELS", 1)
    monkeypatch.setattr(Hh, "K12_MIN_PIXELS", 1)           # the weight gradients of the hidden blocks on K12
    calls = []
    real = Hh.conv3x3
    monkeypatch.setattr(Hh, "conv3x3", lambda x, w, p: (calls.append(tuple(x.shape)), real(x, w, p))[1])
    got, rows, out, p, sd_gpu, sd_cpu = _update_vs_cpu_restatement(1, 128)
    assert len(calls) >= 2 * 4 * 5, "K11 did not run"            # two encoders x 4 optimizer steps x 5 hidden convolutions
    np.testing.assert_allclose(got[0, :6], rows[0, :6], rtol=STEP1_RTOL, atol=STEP1_ATOL)
    _check_later_steps_and_weights(got, rows, out, p, sd_gpu, sd_cpu)


def _winograd_off_worker(_rank, C, S, path):
    res = _update_vs_cpu_restatement(C, S)
    torch.save(dict(got=torch.from_numpy(res[0]), rows=torch.from_numpy(res[1])), path)


@pytest.mark.parametrize("C,S", [(1, 128)], ids=["128x128x1_reference"])
def test_robot_update_step1_with_winograd_off_matches_to_1e5(C, S, tmp_path, monkeypatch):
    """The same comparison in a child process whose MIOpen may not use its Winograd solvers (MIOPEN_DEBUG_CONV_WINOGRAD=0: the
    3x3 convolutions then run as implicit-GEMM fp32 MFMA kernels, a k-ordered fmaf chain like the CPU's direct convolution) --
    the arithmetic closest to the reference's CPU path this library can be asked for.  Step 1 must hold 1e-5 here whatever
    the Winograd transforms do to the product configuration above; a child process because MIOpen reads the switch once."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("MIOPEN_DEBUG_CONV_WINOGRAD", "0")
    path = str(tmp_path / "r.pt")
    mp.start_processes(_winograd_off_worker, args=(C, S, path), nprocs=1, join=True, start_method="spawn")
    r = torch.load(path)
    np.testing.assert_allclose(r["got"][0, :6].numpy(), r["rows"][0, :6].numpy(), rtol=STEP1_RTOL, atol=STEP1_ATOL)
    np.testing.assert_allclose(r["got"][1:, :6].numpy(), r["rows"][1:, :6].numpy(), rtol=LATER_RTOL, atol=LATER_ATOL)


def test_robot_train_runs_on_gpu():
    from aur_ppo_amd.robot_ppo import robot_ppo
    from aur_ppo_amd.robot_run import build_parser, params_from_args
    p = params_from_args(build_parser().parse_args([]))
    p.update(gym_id="Synthetic-arm", num_envs=8, num_steps=16, total_timesteps=2 * 128, num_update_epochs=2,
             num_minibatches=4, pretrain_steps=4, pretrain_batch_size=2, do_pretraining=True, log=False)
    a = robot_ppo(p)
    a.train()
    assert np.isfinite(a._last_scalars).all() and a._last_scalars.shape == (8, 9)


@pytest.mark.parametrize("equivariant", [False, True], ids=["plain_cnn", "equivariant"])
@pytest.mark.parametrize("C,S", [(1, 128), (3, 84)], ids=["config3_width", "config5_shard_width"])
def test_robot_update_at_config_env_count_fused_blocks_equal_stock_blocks(C, S, equivariant):
    """BASELINE configs 3 / 5 run 256 envs per GPU: one update at that env count (T = 2, two minibatches of 256 images), once
    with K9 in the encoder blocks and once with the stock torch ops (``fused_pool = False``), from the same weights, data
    and shuffle seed -- the loss scalars of every step and the final weights must agree (the CPU oracle is hours away at
    this width; at N = 8 both paths are held to it above).  ``equivariant``: the same with the C4-equivariant actor / critic
    BASELINE configs 3 and 5 name (aur_ppo_amd/equiv.py, build-defined: e2cnn is absent; 32 regular
    fields here so that the expanded filter banks stay test-sized) -- K10 / K9 blocks against the stock torch blocks."""
    from aur_ppo_amd.base_cnns import base_encoder
    from aur_ppo_amd.equiv import _Block
    from aur_ppo_amd.robot_ppo import robot_ppo
    from aur_ppo_amd.robot_run import build_parser, params_from_args
    N, T = 256, 2
    p = params_from_args(build_parser().parse_args([]))
    p.update(gym_id="Synthetic-arm", num_envs=N, num_steps=T, total_timesteps=N * T, num_update_epochs=2, num_minibatches=2,
             do_pretraining=False, log=False, obs_size=S, obs_channels=C)
    if equivariant:
        p.update(equivariant=True, equiv_hidden=32)
    outs = []
    for fused in (True, False):
        torch.manual_seed(4)
        a = robot_ppo(p)
        n_seams = 0
        for m in a.policy.modules():
            if isinstance(m, (base_encoder, _Block)):
                m.fused_pool = fused
                n_seams += 1
        assert n_seams > 0
        g = torch.Generator(device="cuda").manual_seed(6)
        b = a.buffer
        b.states.copy_((torch.rand(T, N, device="cuda", generator=g) < 0.5).float())
        b.observations.copy_(torch.rand(T, N, C, S, S, device="cuda", generator=g))
        b.actions.copy_(0.3 * torch.randn(T, N, 5, device="cuda", generator=g))
        b.rewards.copy_((torch.rand(T, N, device="cuda", generator=g) < 0.3).float())
        with torch.no_grad():
            for t in range(T):
                _, _, lp, _, v = a.policy.evaluate(b.states[t], b.observations[t], b.actions[t])
                b.log_probs[t].copy_(lp + 0.05 * torch.randn(N, device="cuda", generator=g))
                b.values[t].copy_(v.flatten())
        a.seed_all(1)
        ret, adv = a.advantages(b.states[0], b.observations[0], torch.zeros(N, device="cuda"), b, T)
        a.update(b.flatten(ret, adv), 2, a.batch_size, a.minibatch_size, [])
        torch.cuda.synchronize()
        outs.append((a._last_scalars.copy(), a.bucket.flat_param.detach().cpu().clone()))
    assert outs[0][0].shape == (4, 9) and np.isfinite(outs[0][0]).all()
    # step 1 runs on identical weights: the two implementations of the blocks (K10's direct first convolution against
    # MIOpen's Winograd one included) must agree to rounding.  From step 2 on Adam has turned every near-zero gradient
    # element's rounding difference into a +-lr step, so the trajectories are compared at that scale.
    np.testing.assert_allclose(outs[0][0][0, :6], outs[1][0][0, :6], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(outs[0][0][:, :6], outs[1][0][:, :6], rtol=5e-3, atol=5e-4)
    d = (outs[0][1] - outs[1][1]).abs()
    assert float(d.max()) <= 1.2e-3 and float((d > 3e-5).float().mean()) < 0.05, (float(d.max()), float((d > 3e-5).float().mean()))


def test_reference_format_checkpoint_rehomes_adam_moments_by_optimizer_order(tmp_path):
    """A reference-format file (src/robot_ppo.py:502-507: actor_state, critic_state, optimizer_state -- torch's per-parameter
    Adam state, numbered in ``policy.parameters()`` order) loaded on the fused-Adam path, whose flat bucket is laid out
    actor first: every parameter's slice of the flat moment buffers must hold THAT parameter's saved moments."""
    from aur_ppo_amd.robot_ppo import robot_ppo
    from aur_ppo_amd.robot_run import build_parser, params_from_args
    p = params_from_args(build_parser().parse_args([]))
    p.update(gym_id="Synthetic-arm", num_envs=2, num_steps=4, total_timesteps=8, num_update_epochs=1, num_minibatches=1,
             do_pretraining=False, log=False, obs_size=128, equivariant=False)
    torch.manual_seed(3)
    a = robot_ppo(p)
    assert a._fused_adam
    opt_order = [q for grp in a.optimizer.param_groups for q in grp["params"]]
    assert [id(q) for q in opt_order] != [id(q) for q in a.bucket.params], "the two orders coincide: the test shows nothing"
    g = torch.Generator().manual_seed(11)
    state = {i: {"step": torch.tensor(7.0), "exp_avg": torch.randn(q.shape, generator=g),
                 "exp_avg_sq": torch.rand(q.shape, generator=g)} for i, q in enumerate(opt_order)}
    osd = {"state": state, "param_groups": [dict(lr=1e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0, amsgrad=False,
                                                 maximize=False, foreach=None, capturable=False, differentiable=False,
                                                 fused=None, params=list(range(len(opt_order))))]}
    path = str(tmp_path / "ref.pt")
    torch.save({"actor_state": {k: v.cpu() for k, v in a.policy.actor.state_dict().items()},
                "critic_state": {k: v.cpu() for k, v in a.policy.critic.state_dict().items()}, "optimizer_state": osd}, path)
    b = robot_ppo(p)
    assert b.load_checkpoint(path) == 0
    pos = {id(q): i for i, q in enumerate(grp_q for grp in b.optimizer.param_groups for grp_q in grp["params"])}
    off = 0
    for q in b.bucket.params:
        k, st = q.numel(), state[pos[id(q)]]
        assert torch.equal(b._adam_m[off:off + k].cpu(), st["exp_avg"].reshape(-1))
        assert torch.equal(b._adam_v[off:off + k].cpu(), st["exp_avg_sq"].reshape(-1))
        # and the torch optimizer's own view of the state is that slice
        assert b.optimizer.state[q]["exp_avg"].data_ptr() == b._adam_m[off:off + k].data_ptr()
        off += k
    assert float(b._adam_t) == 7.0 and abs(b.get_lr() - 1e-4) < 1e-9      # (an fp32 device scalar)
