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


def _worker(rank, world, port, out_dir, over=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    from aur_ppo_amd import dist as D
    from aur_ppo_amd.ppo import ppo
    D.init_from_env(backend="gloo")
    torch.manual_seed(50 + rank)
    agent = ppo(_params(64, **(over or {})))
    assert agent.device.type == "cuda" and agent._mlp is not None and agent.num_envs == 32 and not agent.use_graph
    assert agent._mlp["wide"] == bool(over)
    p0 = agent.bucket.flat_param.clone().cpu()
    sc = _run(agent, _rollout(16, 64), agent.env_lo, agent.env_lo + 32)
    torch.save(dict(p0=p0, p1=agent.bucket.flat_param.clone().cpu(), sc=sc, norms=agent._norms.clone().cpu()),
               os.path.join(out_dir, f"r{rank}.pt"))
    D.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("over", [None, dict(hidden_dim=128, num_layers=3)], ids=["k7_2x64", "k7w_3x128"])
def test_two_ranks_on_the_hip_path(tmp_path, over):
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path), over), nprocs=2, join=True, start_method="spawn")
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
        a = ppo(_params(32, total_timesteps=16 * 32 * 3, hip_graph=False, **(over or {})))
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


# ---------------------------------------------------------------------------------- one-shot exchange over HIP-IPC peer memory
def _p2p_worker(rank, world, port, out_dir, use_graph):
    """AURPPO_DP_ALLREDUCE=p2p: the gradient exchange is csrc/p2p.hip's one-launch all-reduce over buffers the ranks map through
    HIP IPC (same-device handles work, so two ranks on cuda:0 exercise the real protocol: publish, ticket, flag, poll, rank-ordered
    sum).  The gloo group only carries the IPC handles and the initial broadcast."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      AURPPO_DP_ALLREDUCE="p2p", AURPPO_P2P_TIMEOUT_S="20")
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0", "the IPC mode must be in the environment before HIP starts"
    from aur_ppo_amd import dist as D
    from aur_ppo_amd.ppo import ppo
    D.init_from_env(backend="gloo")
    torch.manual_seed(50 + rank)
    agent = ppo(_params(64, hip_graph=use_graph, total_timesteps=16 * 64 * 4))
    assert agent._p2p is not None and agent.num_envs == 32 and agent.use_graph == use_graph and "p2p" in agent.collective
    p0 = agent.bucket.flat_param.clone().cpu()
    sc = _run(agent, _rollout(16, 64), agent.env_lo, agent.env_lo + 32, updates=4)
    status = agent._p2p.status()
    torch.save(dict(p0=p0, p1=agent.bucket.flat_param.clone().cpu(), sc=sc, norms=agent._norms.clone().cpu(), status=status,
                    captured=agent._graph is not None, fallback=agent.graph_fallback, grad=agent.bucket.flat_grad.clone().cpu()),
               os.path.join(out_dir, f"p{rank}_{int(use_graph)}.pt"))
    D.barrier()
    agent._p2p.close()
    torch.distributed.destroy_process_group()


def test_two_ranks_exchange_gradients_over_ipc_peer_memory(tmp_path, monkeypatch):
    """SURVEY 8e's plan B on the real HIP path: 4 updates x 8 optimizer steps of {K7 grad, one-shot exchange, clip + Adam}, eagerly and
    as one hipGraph per update.  Every rank must end on the same bits (the sum is formed in rank order everywhere), no flag may
    time out, and the run must end where the gloo all-reduce of test_two_ranks_on_the_hip_path ends (the same mean gradient; the
    clip's norm is summed in a different order, hence 1e-6 instead of equality)."""
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for use_graph in (False, True):
        mp.start_processes(_p2p_worker, args=(2, _free_port(), str(tmp_path), use_graph), nprocs=2, join=True, start_method="spawn")
    e0, e1 = (torch.load(tmp_path / f"p{k}_0.pt") for k in range(2))
    g0, g1 = (torch.load(tmp_path / f"p{k}_1.pt") for k in range(2))
    for a, b in ((e0, e1), (g0, g1)):
        assert a["status"] == 0 and b["status"] == 0, (a["status"], b["status"])
        assert torch.equal(a["p0"], b["p0"])
        assert torch.equal(a["p1"], b["p1"]) and torch.equal(a["norms"], b["norms"]) and torch.equal(a["grad"], b["grad"])
        assert not torch.equal(a["sc"], b["sc"])                  # per-shard losses
    assert g0["captured"] and g1["captured"], (g0["fallback"], g1["fallback"])
    assert not e0["captured"]
    torch.testing.assert_close(g0["sc"], e0["sc"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(g0["p1"], e0["p1"], rtol=1e-6, atol=1e-8)
    # the same run through the process group's all-reduce (gloo here, RCCL on a multi-GPU node)
    port = _free_port()
    mp.start_processes(_worker_updates, args=(2, port, str(tmp_path), 4), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "r0.pt")
    assert torch.equal(r0["p0"], e0["p0"])
    torch.testing.assert_close(e0["sc"], r0["sc"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(e0["p1"], r0["p1"], rtol=1e-6, atol=1e-7)


def _p2p_late_worker(rank, world, port, out_dir):
    """Rank 1 arrives 4 s late at the first exchange; rank 0 waits at most 1 s per exchange."""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      AURPPO_DP_ALLREDUCE="p2p", AURPPO_P2P_TIMEOUT_S="1" if rank == 0 else "30")
    from aur_ppo_amd import dist as D
    from aur_ppo_amd import hip_ops as H
    D.init_from_env(backend="gloo")
    n = 5000
    x = D.make_p2p_exchange(H, n, torch.device("cuda", 0))
    step = torch.ones(1, device="cuda")                      # exchange number 1
    g = torch.full((n,), float(rank + 1), device="cuda")
    D.barrier()
    if rank == 1:
        time.sleep(4.0)
    t0 = time.perf_counter()
    x.allreduce_mean_(g, n, step, timeout_s=float(os.environ["AURPPO_P2P_TIMEOUT_S"]))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    status = x.status()
    torch.save(dict(status=status, dt=dt, mean=float(g[0])), os.path.join(out_dir, f"late{rank}.pt"))
    D.barrier()
    x.close()
    torch.distributed.destroy_process_group()


def test_exchange_with_a_late_peer_times_out_instead_of_hanging(tmp_path, monkeypatch):
    """The exchange kernel polls its peers' flags with a wall-clock bound: a peer that does not arrive must cost the waiting rank its
    timeout, raise the sticky status (1 + the late rank), and let the grid drain -- not hang the GPU.  The late rank itself finds
    rank 0's flag already raised and completes normally."""
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.start_processes(_p2p_late_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(tmp_path / "late0.pt"), torch.load(tmp_path / "late1.pt")
    assert r0["status"] == 2, r0                      # 1 + rank 1
    assert 0.9 <= r0["dt"] <= 3.5, r0                 # waited its one second, not the peer's four
    assert r1["status"] == 0 and r1["mean"] == 1.5, r1


def _worker_updates(rank, world, port, out_dir, updates):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    os.environ.pop("AURPPO_DP_ALLREDUCE", None)
    from aur_ppo_amd import dist as D
    from aur_ppo_amd.ppo import ppo
    D.init_from_env(backend="gloo")
    torch.manual_seed(50 + rank)
    agent = ppo(_params(64, total_timesteps=16 * 64 * 4))
    assert agent._p2p is None
    p0 = agent.bucket.flat_param.clone().cpu()
    sc = _run(agent, _rollout(16, 64), agent.env_lo, agent.env_lo + 32, updates=updates)
    torch.save(dict(p0=p0, p1=agent.bucket.flat_param.clone().cpu(), sc=sc), os.path.join(out_dir, f"r{rank}.pt"))
    D.barrier()
    torch.distributed.destroy_process_group()


# ---------------------------------------------------------------------------------- RCCL: the captured collective
def _rccl_worker(rank, world, port, out_dir, force_dp, use_graph, over=None):
    """One process per device over backend "nccl" (= RCCL).  world == 1 + force_dp: the multi-GPU launch path (K7 grad ->
    all-reduce -> apply) through a communicator of one -- what a one-GPU box can rehearse of the captured collective."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    torch.distributed.init_process_group("nccl", rank=rank, world_size=world)
    from aur_ppo_amd.ppo import ppo
    torch.manual_seed(50 + rank)
    agent = ppo(_params(64, device=torch.device("cuda", rank), force_dp=force_dp, hip_graph=use_graph,
                        total_timesteps=16 * 64 * 4, **(over or {})))
    per = 64 // world
    assert agent._dp and agent.num_envs == per and agent.use_graph == use_graph
    p0 = agent.bucket.flat_param.clone().cpu()
    sc = _run(agent, _rollout(16, 64), agent.env_lo, agent.env_lo + per, updates=4)
    torch.save(dict(p0=p0, p1=agent.bucket.flat_param.clone().cpu(), sc=sc, norms=agent._norms.clone().cpu(),
                    captured=agent._graph is not None, fallback=agent.graph_fallback),
               os.path.join(out_dir, f"r{rank}_{int(use_graph)}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_one_rank_rccl_update_is_captured_as_hipgraph(tmp_path):
    """The W > 1 update (16 x {aurppo_mlp_ppo_grad_f32, RCCL all-reduce, aurppo_mlp_ppo_apply_f32}) recorded into ONE
    hipGraph and replayed: eager / capture / replay / replay must end where four eager updates end, and where the
    single-process chained path ends."""
    for use_graph in (True, False):
        mp.start_processes(_rccl_worker, args=(1, _free_port(), str(tmp_path), True, use_graph), nprocs=1, join=True,
                           start_method="spawn")
    g, e = torch.load(tmp_path / "r0_1.pt"), torch.load(tmp_path / "r0_0.pt")
    assert g["captured"], f"the update with the collective inside was not captured: {g['fallback']}"
    assert not e["captured"]
    torch.testing.assert_close(g["sc"], e["sc"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(g["p1"], e["p1"], rtol=1e-6, atol=1e-8)
    # the plain single-process trainer (chained K7 -> reduce -> clip+Adam) from the same start
    from aur_ppo_amd.ppo import ppo
    a = ppo(_params(64, total_timesteps=16 * 64 * 4, hip_graph=False))
    with torch.no_grad():
        a.bucket.flat_param.copy_(g["p0"].cuda())
    sc = _run(a, _rollout(16, 64), 0, 64, updates=4)
    torch.testing.assert_close(g["sc"], sc, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(g["p1"], a.bucket.flat_param.cpu(), rtol=1e-5, atol=1e-7)


def test_one_rank_rccl_update_with_a_wide_policy_is_captured(tmp_path):
    """-d 128 -nl 3 on the W > 1 launch path: 16 x {K7w step, RCCL all-reduce, K6b} in one hipGraph; ends where the eager
    run and the plain single-process trainer end."""
    over = dict(hidden_dim=128, num_layers=3)
    for use_graph in (True, False):
        mp.start_processes(_rccl_worker, args=(1, _free_port(), str(tmp_path), True, use_graph, over), nprocs=1, join=True,
                           start_method="spawn")
    g, e = torch.load(tmp_path / "r0_1.pt"), torch.load(tmp_path / "r0_0.pt")
    assert g["captured"], f"the update with the collective inside was not captured: {g['fallback']}"
    torch.testing.assert_close(g["sc"], e["sc"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(g["p1"], e["p1"], rtol=1e-6, atol=1e-8)
    from aur_ppo_amd.ppo import ppo
    a = ppo(_params(64, total_timesteps=16 * 64 * 4, hip_graph=False, **over))
    assert a._mlp["wide"]
    with torch.no_grad():
        a.bucket.flat_param.copy_(g["p0"].cuda())
    sc = _run(a, _rollout(16, 64), 0, 64, updates=4)
    torch.testing.assert_close(g["sc"], sc, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(g["p1"], a.bucket.flat_param.cpu(), rtol=1e-5, atol=1e-7)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_ranks_over_rccl_captured(tmp_path):
    """What test_two_ranks_on_the_hip_path asserts, over backend "nccl" on two devices, with the update captured."""
    mp.start_processes(_rccl_worker, args=(2, _free_port(), str(tmp_path), False, True), nprocs=2, join=True,
                       start_method="spawn")
    r0, r1 = (torch.load(tmp_path / f"r{k}_1.pt") for k in range(2))
    assert r0["captured"] and r1["captured"], (r0["fallback"], r1["fallback"])
    assert torch.equal(r0["p0"], r1["p0"])
    assert torch.equal(r0["p1"], r1["p1"])
    assert torch.equal(r0["norms"], r1["norms"])
    assert not torch.equal(r0["sc"], r1["sc"])
    mp.start_processes(_rccl_worker, args=(2, _free_port(), str(tmp_path), False, False), nprocs=2, join=True,
                       start_method="spawn")
    e0 = torch.load(tmp_path / "r0_0.pt")
    assert not e0["captured"]
    torch.testing.assert_close(r0["p1"], e0["p1"], rtol=1e-6, atol=1e-8)
    torch.testing.assert_close(r0["sc"], e0["sc"], rtol=1e-6, atol=1e-7)
