"""CPU: robot_ppo API surface and host logic with the oracle standing in for the kernels."""
import numpy as np
import torch

from aur_ppo_amd.robot_ppo import robot_ppo, store_returns, torch_buffer
from aur_ppo_amd.robot_run import build_parser, params_from_args
from tests import oracle_ops


def _params(**over):
    p = params_from_args(build_parser().parse_args([]))
    p.update(gym_id="Synthetic-arm", num_envs=2, num_steps=6, total_timesteps=24, num_update_epochs=2, num_minibatches=2,
             pretrain_steps=3, pretrain_batch_size=2, do_pretraining=False, obs_size=128, log=False, device="cpu")
    p.update(over)
    return p


def test_cli_defaults_match_reference():
    p = params_from_args(build_parser().parse_args([]))     # src/robot_run.py:41-84
    assert (p["gym_id"], p["num_steps"], p["total_timesteps"], p["num_update_epochs"], p["learning_rate"], p["num_envs"]) == \
        ("close_loop_block_reaching", 1024, 50000, 10, 3e-4, 5)
    assert (p["pretrain_episodes"], p["pretrain_steps"], p["pretrain_batch_size"], p["expert_weight"]) == (100, 1000, 8, 0.9)
    assert p["continuous"] is True and p["equivariant"] is False and p["do_pretraining"] is True and p["save_file_path"] is None


def test_api_surface_and_buffer_layout():
    a = robot_ppo(_params(), ops=oracle_ops)
    for m in ("rewards_to_go", "expert_rollout", "run_gae", "normal_advantage", "advantages", "pretrain", "pretrain_update",
              "test_env", "update", "train"):
        assert callable(getattr(a, m)), m
    b = a.buffer
    assert isinstance(b, torch_buffer)
    assert b.states.shape == (6, 2) and b.observations.shape == (6, 2, 1, 128, 128) and b.true_actions.shape == (6, 2, 5)
    assert b.log_probs.shape == (6, 2)          # intended (T,N) semantics
    flat = b.flatten(torch.zeros(6, 2), torch.zeros(6, 2))
    assert [tuple(t.shape) for t in flat] == [(12,), (12, 1, 128, 128), (12,), (12, 5), (12,), (12,), (12,), (12, 5)]
    s = store_returns(2, 0.5)
    for r in (1.0, 1.0, 1.0):
        s.add_value(0, r)
    assert s.calc_discounted_return(0) == (1.75, 3) and s.env_returns[0] == []


def test_skip_last_gae_is_the_default_and_can_be_fixed():
    a = robot_ppo(_params(), ops=oracle_ops)
    a.buffer.rewards.fill_(1.0)
    nv, nd = torch.zeros(2), torch.zeros(2)
    ret, adv = a.run_gae(nv, nd, a.buffer, 6)
    np.testing.assert_allclose(adv[:, 0].numpy(), [4.439, 3.657, 2.825, 1.9405, 1.0, 0.0], atol=1e-3)
    a2 = robot_ppo(_params(fix_gae_bootstrap=True), ops=oracle_ops)
    a2.buffer.rewards.fill_(1.0)
    _, adv2 = a2.run_gae(nv, nd, a2.buffer, 6)
    assert adv2[-1, 0] == 1.0


def test_train_end_to_end_with_pretraining_and_update_return_signature():
    torch.manual_seed(0)
    a = robot_ppo(_params(do_pretraining=True), ops=oracle_ops)
    before = a.bucket.flat_param.clone()
    a.train()
    assert not torch.equal(before, a.bucket.flat_param)
    tags = {t for (t, _, _) in a.writer.scalars}
    assert {"charts/learning_rate", "losses/value_loss", "losses/policy_loss", "losses/entropy", "losses/old_approx_kl",
            "losses/approx_kl", "losses/clipfrac", "losses/explained_variance", "charts/SPS"} <= tags
    # update(): 6-tuple, value loss already weighted, clip over the actor slice only
    ret, adv = a.advantages(*a.envs.reset(), torch.zeros(2), a.buffer, a.num_steps)
    out = a.update(a.buffer.flatten(ret, adv), 1, a.batch_size, a.minibatch_size, [])
    assert len(out) == 6 and len(out[5]) == 2
    np.testing.assert_allclose(float(out[1]), a._last_scalars[-1][2] * a.value_coeff, rtol=1e-6)
    assert a.n_actor == sum(p.numel() for p in a.policy.actor.parameters())


def test_pretrain_update_leaves_the_policy_optimizer_untouched():
    """upstream's optimizer.step() in pretrain_update is a no-op (every policy gradient is None), so real
    training starts with Adam's bias corrections at t = 1 -- the optimizer must not have been stepped."""
    torch.manual_seed(0)
    a = robot_ppo(_params(do_pretraining=True, num_update_epochs=2), ops=oracle_ops)
    before = a.bucket.flat_param.clone()
    a.seed_all(1)
    ns, no, nd = a.pretrain()
    ret, adv = a.advantages(ns, no, nd, a.pretrain_buffer, a.pretrain_steps)
    key0, pos0 = a.rng.get_state()
    a.pretrain_update(a.pretrain_buffer.flatten(ret, adv), 2, a.pretrain_batch_size, a.pretrain_minibatch_size)
    assert torch.equal(before, a.bucket.flat_param)
    assert all(len(s) == 0 or float(s["step"]) == 0 for s in a.optimizer.state.values()), "Adam step counter advanced"
    assert a.rng.get_state()[1] != pos0 or not np.array_equal(a.rng.get_state()[0], key0)     # the shuffles were still drawn


def test_build_defined_84x84x3_policy_and_env():
    a = robot_ppo(_params(obs_size=84, obs_channels=3), ops=oracle_ops)
    assert a.buffer.observations.shape == (6, 2, 3, 84, 84) and a.policy.obs_shape == (3, 84, 84)
    s, o = a.envs.reset()
    assert o.shape == (2, 3, 84, 84)
    acts, unscaled, lp, ent, v = a.policy.evaluate(s, o)
    assert acts.shape == (2, 5) and lp.shape == (2,) and v.shape == (2, 1)
    # state plane folded into conv 1 == the materialised concat (the reference's formulation, robot_actor_critic.py:58-59)
    x = torch.cat([o, s.reshape(-1, 1, 1, 1).repeat(1, 1, 84, 84)], dim=1)
    np.testing.assert_allclose(a.policy.critic(o, s).detach().numpy(), a.policy.critic(x).detach().numpy(), rtol=1e-4, atol=1e-5)
    import pytest
    with pytest.raises(ValueError):
        robot_ppo(_params(obs_size=100), ops=oracle_ops)


def test_channels_last_option_gives_the_same_numbers():
    torch.manual_seed(1)
    a = robot_ppo(_params(), ops=oracle_ops)
    s, o = a.envs.reset()
    v0 = a.policy.value(s, o).detach().clone()
    a.policy.memory_format = torch.channels_last
    np.testing.assert_allclose(a.policy.value(s, o).detach().numpy(), v0.numpy(), rtol=1e-4, atol=1e-5)


def test_target_kl_early_stop_leaves_the_generator_where_numpy_would_be():
    """src/robot_ppo.py:338,406-408: upstream shuffles once per epoch it runs; a stop after epoch 1 must have consumed ONE
    shuffle of the global stream, not ``update_epochs`` of them."""
    a = robot_ppo(_params(target_kl=-1.0, num_update_epochs=3), ops=oracle_ops)
    a.seed_all(1)
    ret, adv = a.advantages(*a.envs.reset(), torch.zeros(2), a.buffer, a.num_steps)
    a.update(a.buffer.flatten(ret, adv), 3, a.batch_size, a.minibatch_size, [])
    assert a._last_scalars.shape[0] == 2                      # one epoch x 2 minibatches
    rs = np.random.RandomState(1)
    rs.shuffle(np.arange(a.batch_size))
    key, pos = a.rng.get_state()
    np.testing.assert_array_equal(key, rs.get_state()[1])
    assert pos == rs.get_state()[2]


def test_mid_run_checkpoint_and_resume_through_train(tmp_path):
    """--checkpoint_path / --checkpoint_every write a checkpoint DURING the run; --resume continues from it: the remaining
    updates only, no second pre-training phase (upstream writes one file at the very end and cannot resume,
    src/robot_ppo.py:502-507)."""
    path = str(tmp_path / "mid.pt")
    torch.manual_seed(0)
    a = robot_ppo(_params(total_timesteps=4 * 12, checkpoint_path=path, checkpoint_every=2, do_pretraining=True), ops=oracle_ops)
    assert a.num_updates == 4
    saved = []
    save0 = a.save_checkpoint
    a.save_checkpoint = lambda p, update=0: (saved.append(update), save0(p, update=update))
    a.train()
    assert saved == [2, 4]
    sd = torch.load(path, weights_only=True)            # tensors, numbers, lists and dicts only
    assert sd["update"] == 4 and {"actor_state", "critic_state", "optimizer_state", "trainer_state"} <= set(sd)
    # resume from a checkpoint taken after update 2
    b = robot_ppo(_params(total_timesteps=4 * 12, checkpoint_path=path, checkpoint_every=2, do_pretraining=True), ops=oracle_ops)
    b.save_checkpoint(path, update=2)
    c = robot_ppo(_params(total_timesteps=4 * 12, resume=path, do_pretraining=True), ops=oracle_ops)
    c.pretrain = lambda *a_, **k: (_ for _ in ()).throw(AssertionError("a resumed run must not pre-train again"))
    c.train()
    n_updates = sum(1 for (t, _, _) in c.writer.scalars if t == "losses/policy_loss")
    assert n_updates == 2


def test_reference_format_checkpoint_loads(tmp_path):
    """A file with upstream's three keys only (src/robot_ppo.py:502-507: actor_state, critic_state, optimizer_state)."""
    a = robot_ppo(_params(), ops=oracle_ops)
    ret, adv = a.advantages(*a.envs.reset(), torch.zeros(2), a.buffer, a.num_steps)
    a.update(a.buffer.flatten(ret, adv), 1, a.batch_size, a.minibatch_size, [])          # gives the optimizer a state
    path = str(tmp_path / "ref.pt")
    torch.save({"actor_state": a.policy.actor.state_dict(), "critic_state": a.policy.critic.state_dict(),
                "optimizer_state": a.optimizer.state_dict()}, path)
    b = robot_ppo(_params(), ops=oracle_ops)
    assert b.load_checkpoint(path) == 0
    for (k, x), (_, y) in zip(a.policy.actor.state_dict().items(), b.policy.actor.state_dict().items()):
        assert torch.equal(x, y), k
    sa, sb = a.optimizer.state_dict()["state"], b.optimizer.state_dict()["state"]
    assert sa.keys() == sb.keys() and all(torch.equal(sa[i]["exp_avg"], sb[i]["exp_avg"]) for i in sa)
