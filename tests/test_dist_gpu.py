"""GPU, 2 ranks sharing cuda:0 over gloo: the env-sharded trainer on the real HIP path (K1, K2, K7,
gradient all-reduce, K6b).  RCCL itself needs one device per rank, so the collective here is gloo's
(CUDA tensors staged through the host); everything else is the code the 8-GPU launch runs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _params(num_envs, **over):
    p = dict(gym_id="Synthetic-v0", seed=1.0, num_steps=16, gae=True, total_timesteps=16 * num_envs * 3, anneal_lr=True,
             gae_lambda=0.95, num_update_epochs=2, num_envs=num_envs, num_minibatches=4, entropy_coeff=0.0,
             value_coeff=0.5, clip_coeff=0.2, clip_vloss=True, max_grad_norm=0.5, target_kl=None, norm_adv=True,
             capture_video=False, hidden_dim=64, continuous=True, learning_rate=3e-4, exp_name="t", num_layers=2,
             dropout=0.0, gamma=0.99, track=False, log=False, save=False, obs_dim=16, act_dim=3)
    p.update(over)
    return p


def _rollout(T, N, seed=3):
    rs = np.random.RandomState(seed)
    return dict(states=rs.standard_normal((T, N, 16)).astype(np.float32), actions=rs.standard_normal((T, N, 3)).astype(np.float32),
                log_probs=(-4 + 0.1 * rs.standard_normal((T, N))).astype(np.float32),
                rewards=rs.standard_normal((T, N)).astype(np.float32), values=rs.standard_normal((T, N)).astype(np.float32),
                terminals=(rs.random_sample((T, N)) < 0.05).astype(np.float32),
                next_obs=rs.standard_normal((N, 16)).astype(np.float32), next_done=np.zeros(N, np.float32))


def _run(agent, full, lo, hi, updates=3):
    dev = agent.device
    for k in ("states", "actions", "log_probs", "rewards", "values", "terminals"):
        getattr(agent.buffer, k).copy_(torch.from_numpy(full[k][:, lo:hi]))
    agent.seed_all(1)
    sc = []
    for u in range(updates):
        agent.set_lr((1 - u / updates) * 3e-4)
        ret, adv = agent.advantages(torch.from_numpy(full["next_obs"][lo:hi]).to(dev), torch.from_numpy(full["next_done"][lo:hi]).to(dev))
        n = agent.update(ret, adv)
        sc.append(agent._scalars[:n].clone().cpu())
    torch.cuda.synchronize()
    return torch.stack(sc)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    from aur_ppo_amd import dist as D
    from aur_ppo_amd.ppo import ppo
    D.init_from_env(backend="gloo")
    torch.manual_seed(50 + rank)
    agent = ppo(_params(64))
    assert agent.device.type == "cuda" and agent._mlp is not None and agent.num_envs == 32 and not agent.use_graph
    p0 = agent.bucket.flat_param.clone().cpu()
    sc = _run(agent, _rollout(16, 64), agent.env_lo, agent.env_lo + 32)
    torch.save(dict(p0=p0, p1=agent.bucket.flat_param.clone().cpu(), sc=sc, norms=agent._norms.clone().cpu()),
               os.path.join(out_dir, f"r{rank}.pt"))
    D.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_on_the_hip_path(tmp_path):
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = (torch.load(tmp_path / f"r{k}.pt") for k in range(2))
    assert torch.equal(r0["p0"], r1["p0"])                    # broadcast at construction
    assert torch.equal(r0["p1"], r1["p1"])                    # same reduced gradients -> same weights, bit for bit
    assert torch.equal(r0["norms"], r1["norms"])
    assert not torch.equal(r0["sc"], r1["sc"])                # per-shard losses
    # single-process emulation: two shard trainers stepped in lock-step with their flat gradients averaged
    from aur_ppo_amd.ppo import ppo
    full = _rollout(16, 64)
    agents = []
    for rank in range(2):
        a = ppo(_params(32, total_timesteps=16 * 32 * 3, hip_graph=False))
        with torch.no_grad():
            a.bucket.flat_param.copy_(r0["p0"].cuda())
        agents.append(a)
    # run both shards minibatch by minibatch: K7 on each shard, average the flat gradients, clip + Adam on both
    a0, a1 = agents
    for a, lo in ((a0, 0), (a1, 32)):
        for k in ("states", "actions", "log_probs", "rewards", "values", "terminals"):
            getattr(a.buffer, k).copy_(torch.from_numpy(full[k][:, lo:lo + 32]))
        a.seed_all(1)
    for u in range(3):
        for a, lo in ((a0, 0), (a1, 32)):
            a.set_lr((1 - u / 3) * 3e-4)
            a._ra = a.advantages(torch.from_numpy(full["next_obs"][lo:lo + 32]).cuda(), torch.from_numpy(full["next_done"][lo:lo + 32]).cuda())
            a._perm = a._take_perms()
        B, M = a0.batch_size, a0.minibatch_size
        H = a0.ops
        for ep in range(2):
            for start in range(0, B, M):
                for a in (a0, a1):
                    b = a.buffer.flatten(*a._ra)
                    H.mlp_ppo_step(b[0], b[2], a._rec, a._perm[ep][start:start + M], a.bucket.flat_param, a._mlp,
                                   a.bucket.flat_grad, 0.2, 0.0, 0.5, True, H.VLOSS_CLIPPED)
                mean = (a0.bucket.flat_grad + a1.bucket.flat_grad) / 2
                for a in (a0, a1):
                    a.bucket.flat_grad.copy_(mean)
                    a._clip_and_step(a._norms[:1])
    torch.cuda.synchronize()
    torch.testing.assert_close(a0.bucket.flat_param.cpu(), r0["p1"], rtol=1e-6, atol=1e-7)
