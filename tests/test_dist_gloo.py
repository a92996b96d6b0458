"""CPU, world_size 2 over gloo: env sharding, the flat-bucket gradient all-reduce and the
trainer's multi-rank update (kernels stood in by the oracle) -- the N>1 path the driver runs
over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _params(num_envs):
    return dict(gym_id="Synthetic-v0", seed=1.0, num_steps=16, gae=True, total_timesteps=16 * num_envs * 2,
                anneal_lr=True, gae_lambda=0.95, num_update_epochs=2, num_envs=num_envs, num_minibatches=2,
                entropy_coeff=0.0, value_coeff=0.5, clip_coeff=0.2, clip_vloss=True, max_grad_norm=0.5, target_kl=None,
                norm_adv=True, capture_video=False, hidden_dim=32, continuous=True, learning_rate=3e-4, exp_name="t",
                num_layers=2, dropout=0.0, gamma=0.99, track=False, log=False, save=False, device="cpu", obs_dim=6,
                act_dim=2)


def _rollout(T, N, seed):
    rs = np.random.RandomState(seed)
    return dict(states=rs.standard_normal((T, N, 6)).astype(np.float32), actions=rs.standard_normal((T, N, 2)).astype(np.float32),
                log_probs=(-1 + 0.1 * rs.standard_normal((T, N))).astype(np.float32),
                rewards=rs.standard_normal((T, N)).astype(np.float32), values=rs.standard_normal((T, N)).astype(np.float32),
                terminals=(rs.random_sample((T, N)) < 0.05).astype(np.float32),
                next_obs=rs.standard_normal((N, 6)).astype(np.float32), next_done=np.zeros(N, np.float32))


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from aur_ppo_amd import dist as D
    from aur_ppo_amd.ppo import ppo
    from tests import oracle_ops
    r, lr, w = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and D.world_size() == world
    assert D.shard_envs(8, rank, world) == (4 * rank, 4 * rank + 4)
    torch.manual_seed(100 + rank)                      # different init per rank: broadcast must fix it
    agent = ppo(_params(8), ops=oracle_ops)
    assert agent.num_envs == 4 and agent.global_num_envs == 8 and agent.batch_size == 64
    assert agent.num_updates == 2                      # total_timesteps counts all ranks' env steps
    full = _rollout(16, 8, 7)
    lo, hi = agent.env_lo, agent.env_lo + agent.num_envs
    for k in ("states", "actions", "log_probs", "rewards", "values", "terminals"):
        getattr(agent.buffer, k).copy_(torch.from_numpy(full[k][:, lo:hi]))
    agent.seed_all(1)
    p0 = agent.bucket.flat_param.clone()
    ret, adv = agent.advantages(torch.from_numpy(full["next_obs"][lo:hi]), torch.from_numpy(full["next_done"][lo:hi]))
    n = agent.update(ret, adv)
    torch.save(dict(p0=p0, p1=agent.bucket.flat_param.clone(), scalars=agent._scalars[:n].clone(), adv=adv.clone(),
                    norms=agent._norms[:n].clone()), os.path.join(out_dir, f"r{rank}.pt"))
    # flat-bucket all-reduce in isolation: mean over ranks
    g = torch.full((10,), float(rank + 1))
    D.allreduce_mean_(g)
    assert torch.allclose(g, torch.full((10,), 1.5))
    D.barrier()
    dist.destroy_process_group()


def test_two_rank_env_sharded_update(tmp_path):
    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r0, r1 = (torch.load(tmp_path / f"r{k}.pt") for k in range(2))
    assert torch.equal(r0["p0"], r1["p0"]), "rank 0's parameters must be broadcast at construction"
    assert torch.equal(r0["p1"], r1["p1"]), "identical reduced gradients -> identical parameters on every rank"
    assert torch.equal(r0["norms"], r1["norms"])          # clip norm is taken on the REDUCED gradient
    assert not torch.equal(r0["scalars"], r1["scalars"])  # losses are per-shard (local minibatches)
    assert not torch.equal(r0["p0"], r0["p1"])

    # single-process emulation of the same semantics: per-shard loss gradients, mean, clip, Adam
    from aur_ppo_amd.ppo import ppo
    from tests import oracle_ops
    torch.set_num_threads(1)
    full = _rollout(16, 8, 7)
    agents = []
    for rank in range(2):
        p = _params(8)
        p["num_envs"] = 4                                   # one local shard, no process group
        p["total_timesteps"] = 16 * 4 * 2
        a = ppo(p, ops=oracle_ops)
        with torch.no_grad():
            a.bucket.flat_param.copy_(r0["p0"])
        lo = 4 * rank
        for k in ("states", "actions", "log_probs", "rewards", "values", "terminals"):
            getattr(a.buffer, k).copy_(torch.from_numpy(full[k][:, lo:lo + 4]))
        a.seed_all(1)
        a._shard = (torch.from_numpy(full["next_obs"][lo:lo + 4]), torch.from_numpy(full["next_done"][lo:lo + 4]))
        agents.append(a)
    # invariant on the first optimizer step: the reduced gradient == mean of the per-shard gradients
    if True:
        grads = []
        for a in agents:
            ret, adv = a.advantages(*a._shard)
            np.testing.assert_array_equal(adv.numpy(), (r0 if a is agents[0] else r1)["adv"].numpy())
            b = a.buffer.flatten(ret, adv)
            perm = a._take_perms()[0][:a.minibatch_size]
            mb = oracle_ops.gather(perm, [b[0], b[2], a._rec])
            _, lp, ent, v = a.policy.evaluate(mb[0], mb[1])
            loss = oracle_ops.ppo_loss_packed(lp, v, ent, mb[2], 0.2, 0.0, 0.5, True, 1)
            a.bucket.zero_grad()
            loss.backward()
            grads.append(a.bucket.flat_grad.clone())
        mean_grad = (grads[0] + grads[1]) / 2
        from oracle import c_oracle
        _, norm = c_oracle.grad_norm_clip(mean_grad.numpy(), 0.5)
        np.testing.assert_allclose(float(r0["norms"][0]), norm, rtol=1e-5)


def test_shard_envs_rejects_uneven_split():
    from aur_ppo_amd import dist as D
    with pytest.raises(ValueError):
        D.shard_envs(10, 0, 4)


# ---------------------------------------------------------------------------------- robot_ppo, target_kl, 2 ranks
def _robot_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from aur_ppo_amd import dist as D
    from aur_ppo_amd.robot_ppo import robot_ppo
    from aur_ppo_amd.robot_run import build_parser, params_from_args
    from tests import oracle_ops
    D.init_from_env(backend="gloo")
    p = params_from_args(build_parser().parse_args([]))
    # a threshold only ONE rank's local approx_kl would cross decides nothing: the mean over ranks does
    p.update(gym_id="Synthetic-arm", num_envs=4, num_steps=4, total_timesteps=32, num_update_epochs=3, num_minibatches=2,
             do_pretraining=False, log=False, device="cpu", obs_size=128, target_kl=float(os.environ["AURPPO_TEST_KL"]))
    torch.manual_seed(7 + rank)
    a = robot_ppo(p, ops=oracle_ops)
    assert a.num_envs == 2 and a.world == 2
    g = torch.Generator().manual_seed(11 + rank)
    b = a.buffer
    b.observations.copy_(torch.rand(b.observations.shape, generator=g))
    b.actions.copy_(0.3 * torch.randn(b.actions.shape, generator=g))
    b.rewards.copy_(torch.rand(b.rewards.shape, generator=g))
    b.log_probs.copy_(-4 + (2.0 if rank == 0 else 0.01) * torch.randn(b.log_probs.shape, generator=g))   # rank 0: large KL
    a.seed_all(1)
    ret, adv = a.advantages(torch.zeros(2), torch.rand(2, 1, 128, 128, generator=g), torch.zeros(2), b, 4)
    a.update(b.flatten(ret, adv), 3, a.batch_size, a.minibatch_size, [])
    torch.save(dict(steps=a._last_scalars.shape[0], kl=float(a._last_scalars[-1][5]), p=a.bucket.flat_param.clone()),
               os.path.join(out_dir, f"r{rank}.pt"))
    D.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kl,expect", [("1e9", 6), ("-1", 2)])
def test_robot_ppo_two_ranks_stop_on_the_reduced_kl_together(tmp_path, kl, expect, monkeypatch):
    """Without the KL all-reduce, ranks whose local approx_kl differ leave the epoch loop at different epochs and the
    next gradient all-reduce never completes (a hang).  Both ranks must run the same number of steps."""
    monkeypatch.setenv("AURPPO_TEST_KL", kl)
    mp.start_processes(_robot_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = (torch.load(tmp_path / f"r{k}.pt") for k in range(2))
    assert r0["steps"] == r1["steps"] == expect
    assert torch.equal(r0["p"], r1["p"])
    assert abs(r0["kl"] - r1["kl"]) > 1e-6          # the local values differ; the decision used their mean


def test_allreduce_choice_reads_the_environment(monkeypatch):
    """AURPPO_DP_ALLREDUCE picks the gradient exchange of the MLP policy's bucket: the process group's all-reduce (default) or
    the one-shot exchange over HIP-IPC peer memory; anything else is an error, not a silent default."""
    from aur_ppo_amd import dist as D
    monkeypatch.delenv("AURPPO_DP_ALLREDUCE", raising=False)
    assert D.allreduce_choice() == "rccl"
    monkeypatch.setenv("AURPPO_DP_ALLREDUCE", "P2P")
    assert D.allreduce_choice() == "p2p"
    monkeypatch.setenv("AURPPO_DP_ALLREDUCE", "ring")
    import pytest
    with pytest.raises(ValueError):
        D.allreduce_choice()
