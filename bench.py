#!/usr/bin/env python
"""env-steps/s through GAE + PPO update (BASELINE.json metric) on N MI355X GPUs of one node.

A "step" is one full pass of the hot path over one synthetic rollout batch that is already
resident in HBM: bootstrap value -> K1 GAE -> K2 E epoch shuffles -> E x minibatches of
{K3 gather, policy/value forward, K4+K5 loss fwd+bwd, backward, [RCCL all-reduce], K6 clip,
Adam}.  Workload at --gpus 1 is the configuration the metric is quoted on (num_envs=4096, T=128,
obs 64, act 6, E=4, 4 minibatches, 2x64 tanh MLP); with N GPUs every rank keeps --envs-per-gpu
envs (weak scaling, env-sharded, one gradient all-reduce per optimizer step).

    python bench.py [--gpus N --steps K --warmup W]        (N > 1: spawns one child per GPU itself, before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

The first (untimed) step is also the parity gate: at N=1 its advantages / returns / permutations / 16 x 9 loss scalars /
final weights are compared with the oracle's reference-faithful CPU update of the same tensors (the update the
cpu_baseline leg computes anyway) and the line carries "parity_checked".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 MFMA peak (same guide, matrix cores table)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (same table: ~2.5 PF; never the 2:1-sparsity figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", choices=["mlp", "robot3", "robot5"], default="mlp", help="mlp (default): BASELINE's headline "
                    "metric / config (4096 envs, T 128, MLP policy); robot3 / robot5: robot_ppo's GAE + update on image observations at "
                    "BASELINE config 3's shape (256 envs, T 128, (1,128,128)) / config 5's per-GPU shard (256 envs, T 64, (3,84,84))")
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 100; 10 for the robot workloads)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--num-steps", type=int, default=128)
    ap.add_argument("--obs-dim", type=int, default=64)
    ap.add_argument("--act-dim", type=int, default=6)
    ap.add_argument("--hidden-dim", type=int, default=64, help="-d of src/run_ppo.py:33 (default = the BASELINE workload; other "
                    "shapes run K7w and are not the headline number)")
    ap.add_argument("--num-layers", type=int, default=2, help="-nl of src/run_ppo.py:37")
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--minibatches", type=int, default=4)
    ap.add_argument("--cpu-baseline-updates", type=int, default=3, help="timed CPU-oracle updates, median reported (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU baseline (0 = usable cores, <=16)")
    ap.add_argument("--no-probe", action="store_true", help="do not time the gather kernel with HIP events")
    ap.add_argument("--no-graph", action="store_true", help="run the update eagerly instead of as a hipGraph")
    ap.add_argument("--no-fused-mlp", action="store_true", help="per-op path (K3 + torch nets + K5) instead of K7")
    ap.add_argument("--shard-envs-per-gpu", type=int, default=512, help="second workload measured in the same process: BASELINE "
                    "config 4's per-GPU shard (0 = skip)")
    ap.add_argument("--dp-allreduce", choices=["rccl", "p2p"], default=None, help="gradient exchange between ranks (sets "
                    "AURPPO_DP_ALLREDUCE): the process group's all-reduce, or the one-shot exchange over HIP-IPC peer memory")
    ap.add_argument("--one-exchange", action="store_true", help="at N > 1 do not measure the other gradient exchange as well")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle comparison of the first (untimed) step")
    ap.add_argument("--force-dp", action="store_true", help="one rank, but the multi-GPU launch path: K7 grad -> RCCL "
                    "all-reduce (group of one) -> apply; rehearses the captured collective on a one-GPU box")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 100 if args.workload == "mlp" else 10
    return args


class EventProbe:
    """HIP-event pairs around every launch of the dominant kernel (K3 gather), recorded on the
    stream it is launched on (torch's current stream)."""

    def __init__(self):
        self.pairs = []
        self.on = False

    def begin(self):
        if self.on:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._b = e

    def end(self):
        if self.on:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.pairs.append((self._b, e))

    def mean_ms(self):
        return float(np.mean([b.elapsed_time(e) for b, e in self.pairs])) if self.pairs else None


def hyper(args, world):
    return dict(gym_id="Synthetic-v0", seed=1.0, num_steps=args.num_steps, gae=True,
                total_timesteps=args.num_steps * args.envs_per_gpu * world, anneal_lr=False, gae_lambda=0.95,
                num_update_epochs=args.epochs, num_envs=args.envs_per_gpu * world, num_minibatches=args.minibatches,
                entropy_coeff=0.0, value_coeff=0.5, clip_coeff=0.2, clip_vloss=True, max_grad_norm=0.5,
                target_kl=None, norm_adv=True, capture_video=False, hidden_dim=args.hidden_dim, continuous=True,
                learning_rate=3e-4, exp_name="bench", num_layers=args.num_layers, dropout=0.0, gamma=0.99, track=False,
                log=False, save=False, obs_dim=args.obs_dim, act_dim=args.act_dim)


def synth_buffers(T, N, D, A, seed):
    """SURVEY section 8d synthetic rollout tensors (CPU generator so the CPU baseline sees the same data)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    return dict(states=torch.randn(T, N, D, generator=g), actions=torch.randn(T, N, A, generator=g),
                values=torch.randn(T, N, generator=g), rewards=torch.randn(T, N, generator=g),
                terminals=(torch.rand(T, N, generator=g) < 0.02).float(),
                next_obs=torch.randn(N, D, generator=g), next_done=(torch.rand(N, generator=g) < 0.02).float())


def usable_cores(cap=16):
    """Host threads this process may really use: affinity mask, cgroup CPU quota, and the GPU box's
    per-GPU CPU share (16) -- oversubscribing a quota'd cgroup makes the baseline meaninglessly slow."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, cap))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def c_twins(args, data):
    """the plain-C `_cpu` twins of the non-network kernels (oracle/aurppo_oracle.c: GAE scan, numpy-legacy
    shuffle, gather, loss forward+backward), one thread, timed at the benchmarked sizes: what one update spends in them."""
    from oracle import c_oracle as CO
    T, N = args.num_steps, args.envs_per_gpu
    B, M, E = T * N, T * N // args.minibatches, args.epochs
    f = lambda k: data[k].numpy()
    nv = np.zeros(N, np.float32)

    def best(fn, n=3):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3

    gae_ms = best(lambda: CO.gae(f("rewards"), f("values"), f("terminals"), nv, f("next_done"), 0.99, 0.95))
    mt = CO.MT(1)
    idx = np.arange(B, dtype=np.int32)
    shuf_ms = best(lambda: [mt.shuffle(idx) for _ in range(E)])
    obs = f("states").reshape(B, -1)
    rec = [f(k).reshape(-1) for k in ("log_probs", "rewards", "values", "values")]
    mb = idx[:M].copy()
    gat_ms = best(lambda: [CO.gather(mb, obs)] + [CO.gather(mb, r) for r in rec]) * E * args.minibatches
    a = [r[mb] for r in rec]
    loss_ms = best(lambda: CO.ppo_loss(a[0] + 0.01, a[0], a[1], a[2] + 0.1, a[2], a[3], a[1], 0.2, 0.0, 0.5, True, 1)) * E * args.minibatches
    tot = gae_ms + shuf_ms + gat_ms + loss_ms
    return {"what": "oracle/aurppo_oracle.c (gcc -O2, 1 thread), non-network stages of one update at the benchmarked size",
            "gae_ms": round(gae_ms, 2), f"shuffle_{E}_epochs_ms": round(shuf_ms, 2),
            f"gather_{E * args.minibatches}_minibatches_ms": round(gat_ms, 2),
            f"loss_fwd_bwd_{E * args.minibatches}_minibatches_ms": round(loss_ms, 2), "total_ms_per_update": round(tot, 2),
            "env_steps_per_s_non_network": round(B / (tot * 1e-3), 1)}


def cpu_baseline(args, data, init_sd, n_updates, gpu_first):
    """The oracle's reference-faithful CPU update (same op sequence as src/ppo.py:125-142,213-269)
    timed on this box's host cores -- a reported baseline, never the measured product path.  BASELINE.md section 3
    protocol: 1 warm-up + 3 timed updates, median.  The warm-up update starts from the same weights, data and shuffle
    seed as the GPU's first step: ``gpu_first`` (that step's results) is checked against it -> (baseline, parity)."""
    from oracle import ppo_oracle as O
    cores = args.cpu_threads or usable_cores()
    torch.set_num_threads(cores)
    hp = hyper(args, 1)
    T, N = args.num_steps, args.envs_per_gpu
    net = O.make_actor_critic(args.obs_dim, (args.act_dim,), args.hidden_dim, args.num_layers, True)
    net.load_state_dict(init_sd)
    opt = torch.optim.Adam(net.parameters(), lr=hp["learning_rate"], eps=1e-5)
    buf = {k: data[k] for k in ("states", "actions", "log_probs", "rewards", "terminals", "values")}
    rng = np.random.RandomState(1)
    t0 = time.perf_counter()
    res = O.reference_update(net, opt, buf, data["next_obs"], data["next_done"], hp, rng, collect=True)   # warm-up
    warm = time.perf_counter() - t0
    log(f"cpu baseline warm-up update: {warm:.2f} s on {cores} threads")
    parity = check_parity(gpu_first, res, net) if gpu_first is not None else None
    if warm > 20.0:          # keep the default run bounded: the warm-up itself is the sample
        per, note = warm, "1 full update (cold; longer than the 20 s budget so used as the sample)"
    else:
        ts = []
        for _ in range(n_updates):
            t0 = time.perf_counter()
            O.reference_update(net, opt, buf, data["next_obs"], data["next_done"], hp, rng, collect=False)
            ts.append(time.perf_counter() - t0)
        per = float(np.median(ts))
        note = f"1 warm-up + {n_updates} timed full updates, median"
    out = {"value": T * N / per, "unit": "env-steps/s", "cores": cores, "kind": "port",
           "sample": f"{note} (N={N}, T={T}, E={args.epochs}, {args.minibatches} minibatches) of "
                     f"oracle.reference_update, torch CPU fp32, {cores} threads, {per:.2f} s/update"}
    try:
        out["c_twins"] = c_twins(args, data)
    except Exception as e:      # the twins are an extra figure; never lose the line over them
        out["c_twins"] = {"error": f"{type(e).__name__}: {e}"}
    return out, parity


def check_parity(gpu, res, net):
    """First GPU step vs the oracle's update of the same tensors (tolerances of tests/test_parity_fullsize.py).
    Returns {"ok": bool, ...max deviations...}."""
    out = {}
    try:
        out["perms_bit_exact"] = bool(all(np.array_equal(gpu["perms"][e], res["perms"][e]) for e in range(len(res["perms"]))))
        out["adv_max_abs_err"] = float(np.abs(gpu["adv"] - res["advantages"].numpy()).max())
        out["ret_max_abs_err"] = float(np.abs(gpu["ret"] - res["returns"].numpy()).max())
        cols = [0, 1, 2, 3, 4, 5, 7, 8]
        g, r = gpu["scalars"][:, cols], res["scalars"][:, cols]
        out["scalars_max_excess"] = float((np.abs(g - r) - (1e-5 + 1e-4 * np.abs(r))).max())     # <= 0 passes
        out["clipfrac_max_abs_err"] = float(np.abs(gpu["scalars"][:, 6] - res["scalars"][:, 6]).max())
        # Final weights after the 16 clip + Adam steps.  K7 / K7w hand row tiles to workgroups through a counter, so a
        # gradient element's last bits depend on the launch (two launches on the same inputs: 1.4-7.5e-9 absolute,
        # profiles/r02/grad_repeat_3x64.txt; with AURPPO_STATIC_TILES they are bit-identical, tests/test_determinism.py --
        # summation order, not a race).  A weight whose gradient is far below Adam's eps moves by lr * m / eps per step, a
        # gain of lr / eps = 30 on that difference, and from the second step on the differing weights feed back into the
        # gradients, so the spread grows over the 16 steps instead of adding up linearly (16 x 30 x 7.5e-9 = 3.6e-6 would be
        # the linear figure; observed between two correct runs: <= 3e-8 with K7, up to 9.4e-6 with K7w's both-net workgroups,
        # profiles/r02/repeat_*.txt).  Gate: 2e-6 absolute for K7 (what its repeatability supports, and the K7 tests hold),
        # 2e-5 for the K7w shapes; with static tiles K7w is held to 2e-6 too (tests/test_parity_fullsize.py).  The nets wider than
        # the fused steps cover (per-op path: library GEMMs or k_linear against the oracle's MKL sums) get the same 2e-5: three 256-wide
        # layers end 5.1e-6 from the oracle on either product, every scalar of every step inside 1e-5.
        w_atol = 2e-5 if gpu.get("wide") else 2e-6
        w_ex = []
        for k, v in net.state_dict().items():
            a, b = gpu["weights"][k], v.numpy()
            w_ex.append(float((np.abs(a - b) - (w_atol + 1e-4 * np.abs(b))).max()))
        out["weights_max_excess"] = max(w_ex)
        out["weights_atol"] = w_atol
        M = gpu["minibatch"]
        out["ok"] = bool(out["perms_bit_exact"] and out["adv_max_abs_err"] <= 1e-5 and out["ret_max_abs_err"] <= 1e-5
                         and gpu["scalars"].shape == res["scalars"].shape and out["scalars_max_excess"] <= 0
                         and out["clipfrac_max_abs_err"] <= 1.5 / M and out["weights_max_excess"] <= 0)
    except Exception as e:
        out["ok"] = False
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def check_parity_sharded(args, hp, N, world, init_sd, gpu, own_log_probs):
    """SURVEY 8e's per-shard parity, run on rank 0 when W > 1: its advantages / returns / permutations and the loss scalars
    of optimizer step 1 against ``oracle.reference_update`` on ITS shard, and the gradient the all-reduce delivered for step 1
    against the mean of the per-shard oracle gradients (every shard's synthetic rollout is regenerated here from its seed,
    1234 + rank; step 1 is the step all of whose inputs are known without running the other ranks' trajectories)."""
    from oracle import ppo_oracle as O
    out = {"ranks": world}
    try:
        T, Dm, A = args.num_steps, args.obs_dim, args.act_dim
        hp1 = dict(hp)
        grads, first = [], None
        for r in range(world):
            data = synth_buffers(T, N, Dm, A, 1234 + r)
            net = O.make_actor_critic(Dm, (A,), args.hidden_dim, args.num_layers, True)
            net.load_state_dict(init_sd)
            if r == 0:
                data["log_probs"] = own_log_probs
            else:
                with torch.no_grad():
                    _, lp, _, _ = net.evaluate(data["states"].view(-1, Dm), data["actions"].view(-1, A))
                data["log_probs"] = lp.view(T, N)
            buf = {k: data[k] for k in ("states", "actions", "log_probs", "rewards", "terminals", "values")}
            opt = torch.optim.SGD(net.parameters(), lr=0.0)      # step 1 only: the optimizer never matters
            g = []
            res = O.reference_update(net, opt, buf, data["next_obs"], data["next_done"], hp1, np.random.RandomState(1),
                                     collect=True, stop_after=1, grads_out=g)
            grads.append(g[0].numpy().astype(np.float64))
            if r == 0:
                first = res
        ref_g = np.mean(grads, axis=0)
        B = T * N
        perm = np.arange(B)
        rs = np.random.RandomState(1)
        perms_ok = True
        for e in range(args.epochs):
            rs.shuffle(perm)
            perms_ok = perms_ok and bool(np.array_equal(gpu["perms"][e], perm))
        out["perms_bit_exact"] = perms_ok
        out["adv_max_abs_err"] = float(np.abs(gpu["adv"] - first["advantages"].numpy()).max())
        out["ret_max_abs_err"] = float(np.abs(gpu["ret"] - first["returns"].numpy()).max())
        cols = [0, 1, 2, 3, 4, 5, 7, 8]
        g0, r0 = gpu["scalars"][0, cols], first["scalars"][0, cols]
        out["step1_scalars_max_excess"] = float((np.abs(g0 - r0) - (1e-5 + 1e-4 * np.abs(r0))).max())
        got_g = gpu["first_grad"][:ref_g.size].astype(np.float64)
        out["step1_reduced_grad_max_abs_err"] = float(np.abs(got_g - ref_g).max())
        out["step1_reduced_grad_scale"] = float(np.abs(ref_g).max())
        out["ok"] = bool(perms_ok and out["adv_max_abs_err"] <= 1e-5 and out["ret_max_abs_err"] <= 1e-5
                         and out["step1_scalars_max_excess"] <= 0
                         and out["step1_reduced_grad_max_abs_err"] <= 2e-5 * out["step1_reduced_grad_scale"] + 1e-8)
    except Exception as e:
        out["ok"] = False
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def self_launch(args):
    """`python bench.py --gpus N` as typed: one child per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its
    environment), started BEFORE this process makes any GPU call -- a process that has touched the GPU is never
    re-exec'ed.  Rank 0 prints the JSON line; this parent only waits and relays the exit code."""
    import socket
    import subprocess
    n = args.gpus
    if os.environ.get("AURPPO_BENCH_REHEARSE") != "1" and torch.cuda.device_count() < n:     # device_count() creates no context
        sys.exit(f"bench.py: --gpus {n} but {torch.cuda.device_count()} GPU(s) visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:          # one rank failed: the others would wait in a collective for ever
                    q.terminate()
    sys.exit(rc)


def run_workload(args, rank, world, dev, envs_per_gpu, steps, warmup, with_probe):
    """One workload through the product path: build the trainer with ``envs_per_gpu`` envs on this rank, load the SURVEY 8d
    synthetic rollout (seed 1234 + rank), run the untimed parity step, ``warmup`` steps, then EXACTLY ``steps`` timed steps
    between barrier + synchronize pairs; max over ranks.  Returns everything the JSON line is built from."""
    from aur_ppo_amd import dist as D
    from aur_ppo_amd.ppo import ppo
    torch.manual_seed(1)
    a2 = argparse.Namespace(**vars(args))
    a2.envs_per_gpu = envs_per_gpu
    hp = hyper(a2, world)
    hp["device"] = dev
    hp["fused_mlp"] = not args.no_fused_mlp
    hp["force_dp"] = args.force_dp
    agent = ppo(hp)
    T, N, Dm, A = args.num_steps, agent.num_envs, args.obs_dim, args.act_dim
    data = synth_buffers(T, N, Dm, A, 1234 + rank)
    init_sd = {k: v.detach().cpu().clone() for k, v in agent.policy.state_dict().items()}
    for k in ("states", "actions", "values", "rewards", "terminals"):
        getattr(agent.buffer, k).copy_(data[k])
    next_obs, next_done = data["next_obs"].to(dev), data["next_done"].to(dev)
    with torch.no_grad():   # old log-probs = the policy's own, at init weights (ratio ~ 1 at epoch 0)
        _, lp, _, _ = agent.policy.evaluate(agent.buffer.states.view(-1, Dm), agent.buffer.actions.view(-1, A))
        agent.buffer.log_probs.copy_(lp.view(T, N))
    data["log_probs"] = agent.buffer.log_probs.cpu()
    agent.seed_all(1)
    probe = EventProbe()
    if args.no_graph:
        agent.use_graph = False
    step_no = [0]
    mlp_events = []
    every = 5 if steps < 50 else 10          # the stand-alone probe launch: every 5th step of a short run, every 10th otherwise

    def one_step():
        returns, advantages = agent.advantages(next_obs, next_done)
        agent.update(returns, advantages)
        step_no[0] += 1
        if with_probe and probe.on and not args.no_probe and step_no[0] % every == 1:
            # The update replays as a hipGraph, whose kernels cannot carry readable events: every `every`-th
            # step, time ONE extra stand-alone launch of the dominant kernel on the update's own inputs
            # (it sits inside the timed region: one more ~125 us launch per five / ten 2.2 ms steps, 1.1 % / 0.6 % of `value`).
            if agent._mlp is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                agent.probe_mlp_step(ev)       # events recorded inside the library, around the step kernel only
                mlp_events.append(ev)
            else:
                agent.probe_gather(probe)

    log(f"rank {rank}/{world}: {envs_per_gpu} envs/GPU: setup done; parity step, then {warmup} warm-up steps")
    # step 0 (untimed, eager): the update the oracle's first CPU update is compared with
    dp = world > 1 or args.force_dp
    if dp:
        agent.first_grad_probe = []
    if getattr(agent, "_p2p", None) is not None:
        D.barrier()                      # the one-shot exchange polls its peers against a wall-clock bound: start the first update together
    returns, advantages = agent.advantages(next_obs, next_done)
    n0 = agent.update(returns, advantages)
    torch.cuda.synchronize()
    if getattr(agent, "_p2p", None) is not None:
        # a peer whose flag never arrived (peer memory not visible across these devices, a dead rank) has cost every exchange of this
        # update its timeout and raised the sticky status: agree on it across ranks and give the arm up HERE, together -- not after
        # warmup + steps more updates at 16 timeouts each
        st = torch.tensor([agent._p2p.status()], device=dev, dtype=torch.int32)
        if world > 1:
            torch.distributed.all_reduce(st, op=torch.distributed.ReduceOp.MAX)
        if int(st) != 0:
            raise RuntimeError(f"one-shot exchange: a peer's flag timed out during the first update (status {int(st)}: 1 + the late rank)")
    first_grad = agent.first_grad_probe[0].cpu().numpy() if (dp and agent.first_grad_probe) else None
    agent.first_grad_probe = None
    gpu_first = None
    if rank == 0:
        gpu_first = dict(adv=advantages.cpu().numpy(), ret=returns.cpu().numpy(), perms=agent._last_perms.cpu().numpy(),
                         scalars=agent._scalars[:n0].cpu().numpy().astype(np.float64), minibatch=agent.minibatch_size,
                         wide=bool((agent._mlp is not None and agent._mlp.get("wide")) or
                                   (agent._mlp is None and (args.hidden_dim, args.num_layers) != (64, 2))), first_grad=first_grad,
                         weights={k: v.detach().cpu().numpy().copy() for k, v in agent.policy.state_dict().items()})
    for _ in range(warmup):
        one_step()
    torch.cuda.synchronize()
    log(f"{envs_per_gpu} envs/GPU: warm-up done, timing {steps} steps")
    probe.on = True
    side = []
    agent._perm_events = [] if agent._perm_stream is not None else None
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_ms = []
    for _ in range(steps):
        th = time.perf_counter()
        one_step()
        host_ms.append((time.perf_counter() - th) * 1e3)
        if agent._perm_events is not None:          # when did the main stream finish this update?
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            side.append(e)
    torch.cuda.synchronize()
    D.barrier()
    dt = time.perf_counter() - t0
    probe.on = False
    perm_events, agent._perm_events = agent._perm_events, None
    log(f"{envs_per_gpu} envs/GPU: timed region {dt:.3f} s for {steps} steps")
    if os.environ.get("AURPPO_BENCH_STEPTIMES"):     # diagnostic: how long the host took to ENQUEUE each update (nothing waits in there)
        log("host enqueue time per update (ms): " + " ".join(f"{x:.3f}" for x in host_ms))
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    sc = agent._scalars.cpu().numpy()
    assert np.isfinite(sc).all(), "non-finite loss scalars in the timed region"
    return dict(agent=agent, hp=hp, N=N, dt=dt, data=data, init_sd=init_sd, gpu_first=gpu_first, probe=probe,
                mlp_events=mlp_events, side=side, perm_events=perm_events, every=every)


def robot_small_sample(args, C, S):
    """Parity gate + CPU baseline of the robot workloads on a BOUNDED sample: one ``robot_ppo.update`` at T = 16 and N = 32 (config 3)
    / 128 (config 5) envs (the run's E epochs x 4 minibatches) on the HIP path and through ``oracle.reference_robot_update`` (the
    restatement of src/robot_ppo.py:329-408 on torch-CPU convolutions) from the same weights, data and shuffle seed; the oracle's
    pass is the timed CPU baseline.  Optimizer step 1 runs on identical weights and is held to 1e-5: that is the parity statement.
    From step 2 on Adam has turned every rounding-level gradient element into a +-lr step of either sign (tests/test_robot_gpu.py
    has the derivation; there 4 steps stay within 2e-4), so two correct implementations drift apart: the later steps' loss, policy
    loss, value loss and entropy are only held to a drift bound of 5e-3 (a wrong gradient shows up orders above that), and the
    largest relative difference per scalar is reported."""
    from aur_ppo_amd.robot_actor_critic import robot_actor_critic
    from aur_ppo_amd.robot_ppo import robot_ppo
    from aur_ppo_amd.robot_run import build_parser, params_from_args
    from oracle import ppo_oracle as O
    T, N, E = 16, (32 if S == 128 else 128), args.epochs
    p = params_from_args(build_parser().parse_args([]))
    p.update(gym_id="Synthetic-arm", num_envs=N, num_steps=T, total_timesteps=N * T * 2, num_update_epochs=E, num_minibatches=4,
             do_pretraining=False, log=False, clip_vloss=True, entropy_coeff=0.01, obs_size=S, obs_channels=C)
    torch.manual_seed(2)
    agent = robot_ppo(p)
    cpu = robot_actor_critic(torch.device("cpu"), False, obs_shape=(C, S, S))
    cpu.load_state_dict({k: v.cpu() for k, v in agent.policy.state_dict().items()})
    g = torch.Generator().manual_seed(9)
    buf = dict(states=(torch.rand(T, N, generator=g) < 0.5).float(), observations=torch.rand(T, N, C, S, S, generator=g),
               actions=0.3 * torch.randn(T, N, 5, generator=g), true_actions=torch.zeros(T, N, 5),
               rewards=(torch.rand(T, N, generator=g) < 0.3).float(), terminals=(torch.rand(T, N, generator=g) < 0.1).float())
    with torch.no_grad():
        _, _, lp, _, v = cpu.evaluate(buf["states"].view(-1), buf["observations"].view(-1, C, S, S), buf["actions"].view(-1, 5))
    buf["log_probs"] = lp.view(T, N) + 0.05 * torch.randn(T, N, generator=g)
    buf["values"] = v.view(T, N).clone()
    for k, t in buf.items():
        getattr(agent.buffer, k).copy_(t)
    next_state, next_obs = (torch.rand(N, generator=g) < 0.5).float(), torch.rand(N, C, S, S, generator=g)
    next_done = torch.zeros(N)
    agent.seed_all(1)
    ret, adv = agent.advantages(next_state.cuda(), next_obs.cuda(), next_done.cuda(), agent.buffer, T)
    agent.update(agent.buffer.flatten(ret, adv), E, agent.batch_size, agent.minibatch_size, [])
    got = agent._last_scalars.copy()
    threads = args.cpu_threads or usable_cores()
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    with torch.no_grad():
        nv = cpu.value(next_state, next_obs).flatten()
    ret_o, adv_o = O.gae(buf["rewards"].numpy(), buf["values"].numpy(), buf["terminals"].numpy(), nv.numpy(), next_done.numpy(),
                         0.99, 0.95, O.GAE_MODE_SKIP_LAST)
    flat_cpu = (buf["states"].view(-1), buf["observations"].view(-1, C, S, S), buf["log_probs"].reshape(-1),
                buf["actions"].view(-1, 5), torch.from_numpy(adv_o).reshape(-1), torch.from_numpy(ret_o).reshape(-1),
                buf["values"].reshape(-1), buf["true_actions"].view(-1, 5))
    opt = torch.optim.Adam(cpu.parameters(), lr=p["learning_rate"], eps=1e-5)
    rows = O.reference_robot_update(cpu, opt, flat_cpu, p, np.random.RandomState(1), agent.minibatch_size)
    cpu_s = time.perf_counter() - t0
    adv_err = float(np.abs(adv.cpu().numpy() - adv_o).max())
    s1 = np.abs(got[0, :6] - rows[0, :6]) - (1e-5 * np.abs(rows[0, :6]) + 1e-6)
    later = np.abs(got[1:, :4] - rows[1:, :4]) - (5e-3 * np.abs(rows[1:, :4]) + 2e-3)      # (the policy loss sits near zero)
    rel = (np.abs(got[1:, :6] - rows[1:, :6]) / (np.abs(rows[1:, :6]) + 1e-3)).max(axis=0)
    parity = {"adv_max_abs_err": adv_err, "step1_scalars_max_excess_over_1e-5": float(s1.max()),
              "later_steps_max_excess_over_drift_bound_5e-3": float(later.max()),
              "later_steps_max_rel_diff_per_scalar[loss,pg,vl,ent,old_kl,kl]": [float(f"{x:.3g}") for x in rel],
              "optimizer_steps": int(got.shape[0]), "ok": bool(adv_err <= 1e-5 and s1.max() <= 0 and later.max() <= 0)}
    base = {"value": N * T / cpu_s, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"oracle.reference_robot_update (src/robot_ppo.py:329-408 on torch-CPU convolutions) + skip-last GAE at N={N} envs, "
                      f"T={T}, E={E}, 4 minibatches of {agent.minibatch_size}, obs ({C},{S},{S}): {cpu_s:.1f} s for {N * T} env-steps"}
    del agent
    torch.cuda.empty_cache()
    return base, parity


def main_robot(args, json_fd):
    """``--workload robot3 | robot5``: robot_ppo's GAE + update (src/robot_ppo.py:224-244,329-408) on synthetic image rollouts
    resident in HBM; one step = one whole update (E epochs x minibatches of the CNN actor-critic).  MIOpen compiles its kernels
    on first use of a shape (minutes on a fresh box): all of that happens in the untimed warm-up."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_robot as BR
    cfg = 3 if args.workload == "robot3" else 5
    C, S = (1, 128) if cfg == 3 else (3, 84)
    base = parity = None
    if args.cpu_baseline_updates > 0 and not args.no_parity:
        base, parity = robot_small_sample(args, C, S)
        log(f"small-sample parity: {parity}")
    res = BR.run(cfg, epochs=args.epochs, minibatches=args.minibatches, updates=args.steps, warmup=max(1, args.warmup),
                 kernel_table=True)
    res.pop("_agent", None)
    kt = res.get("kernel_table") or {}
    N, T = res["N"], res["T"]
    ms = res["ms_per_update"]
    flops = res["conv_flops_per_update"]
    roofline = {"bound": "mfma",
                "kernel": "the convolution kernels of the actor / critic encoders, in aggregate: K11 / K12 (csrc/conv.hip: forward, input and "
                          "weight gradients of the hidden blocks as fp32-equivalent products on bf16 MFMAs) and what stays with MIOpen (the "
                          "16 -> 32 block's gradients, the 3x3 -> 1x1 tail, its layout transposes) -- `hand_written_conv_ms` / `miopen_conv_ms`; "
                          "K9 / K10 (csrc/pool.hip) are HBM-bound and listed in `top`",
                "achieved": kt.get("library_conv_tflops"), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(kt["library_conv_tflops"] / MFMA_F32_PEAK_TFLOPS, 4) if kt.get("library_conv_tflops") else None,
                "traffic": None,
                "algorithmic_conv_flops_per_update": flops,
                "whole_update_frac_of_fp32_mfma_peak": res["frac_of_fp32_mfma_peak"],
                "conv_share_of_gpu_time": kt.get("library_conv_share"), "hand_written_conv_ms": kt.get("hand_written_conv_ms"),
                "miopen_conv_ms": kt.get("miopen_conv_ms"), "top": kt.get("top"),
                "how": "direct-convolution FLOPs of both encoders, forward + both gradients (tools/bench_robot.py::conv_flops), over the "
                       "summed device time of every convolution kernel in one more steady-state update under torch.profiler in this "
                       "process; priced at the fp32 MFMA peak the same FLOPs would need (K11 / K12 issue six bf16 products per fp32 product: "
                       "their own pipe's ceiling is 2 500 / 6 = 417 TFLOP/s; Winograd does fewer multiplies than it is credited)"}
    out = {"metric": "env-steps/sec through GAE+PPO-update at num_envs=4096,T=128; 1/2/4/8 GPU",
           "value": N * T / (ms * 1e-3), "unit": "env-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": max(1, args.warmup),
           "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": res["workload"] + f" -- BASELINE config {cfg}" + (" (per-GPU shard of 2048 envs / 8 GPUs)" if cfg == 5 else "")
                                  + ", plain-CNN robot_actor_critic, NOT the headline configuration",
                      "policy": res["policy"], "minibatch_step": "K3 gather + K10 first block + K11 / K12 hidden convolutions (MIOpen: the 16 -> 32 block's gradients, the 1x1 tail) + K9 block tails + K5 loss + K6b"},
           "roofline": roofline, "cpu_baseline": base, "parity_checked": bool(parity["ok"]) if parity else False,
           "parity": parity if parity else "not run (--cpu-baseline-updates 0 / --no-parity)", "graph_fallback": None}
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if parity is not None and not parity["ok"]:
        sys.exit("bench.py: the robot update's small-sample parity gate failed -- see \"parity\" in the line above")


def oracle_full_update(args, hp, N, data, init_sd):
    """oracle.reference_update of one whole update on ``data`` from ``init_sd`` (the checker; CPU)."""
    from oracle import ppo_oracle as O
    net = O.make_actor_critic(args.obs_dim, (args.act_dim,), args.hidden_dim, args.num_layers, True)
    net.load_state_dict(init_sd)
    opt = torch.optim.Adam(net.parameters(), lr=hp["learning_rate"], eps=1e-5)
    buf = {k: data[k] for k in ("states", "actions", "log_probs", "rewards", "terminals", "values")}
    res = O.reference_update(net, opt, buf, data["next_obs"], data["next_done"], hp, np.random.RandomState(1), collect=True)
    return res, net


def main():
    args = parse()
    if args.dp_allreduce:
        os.environ["AURPPO_DP_ALLREDUCE"] = args.dp_allreduce       # (before the ranks are spawned: they inherit it)
    # the one-shot exchange's poll bound while benchmarking: 2 s (the ranks enter every update together here; the trainer's default
    # of 10 s would turn a broken peer mapping into 16 x 10 s per update)
    os.environ.setdefault("AURPPO_P2P_TIMEOUT_S", "2")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    # stdout carries exactly ONE line, the JSON: anything a library prints there (RCCL announces its version on stdout when
    # the first communicator is built) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.workload != "mlp":
        if args.gpus != 1:
            sys.exit("bench.py: the robot workloads are single-GPU lines (config 5's per-GPU shard at N = 1)")
        torch.cuda.set_device(0)
        return main_robot(args, json_fd)
    from aur_ppo_amd import dist as D
    # AURPPO_BENCH_REHEARSE=1: every rank on cuda:0 over gloo -- a rehearsal of the N > 1 code path on a one-GPU box
    # (its timings mean nothing); the driver's runs use one device per rank over RCCL
    rehearse = os.environ.get("AURPPO_BENCH_REHEARSE") == "1"
    if args.force_dp and args.gpus == 1 and "WORLD_SIZE" not in os.environ:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"))
        torch.cuda.set_device(0)
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)      # RCCL communicator of one
    rank, local_rank, world = D.init_from_env(backend="gloo" if rehearse else None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    run = run_workload(args, rank, world, dev, args.envs_per_gpu, args.steps, args.warmup, True)
    # BASELINE config 4 / SURVEY 8e define the scaling curve at 512 envs per GPU: the same step at that shard size, in this
    # same process (every N, so the driver can form the ratio from its N = 1 line)
    shard = None
    if args.shard_envs_per_gpu > 0 and args.shard_envs_per_gpu != args.envs_per_gpu:
        shard = run_workload(args, rank, world, dev, args.shard_envs_per_gpu, args.steps, args.warmup, False)
    # One process per GPU: the same two workloads once more through the OTHER gradient exchange (RCCL's all-reduce <-> the one-shot
    # exchange over HIP-IPC peer memory), in this same process -- so that a single multi-GPU run of the default command says what
    # either costs.  `value` stays the primary arm's; this is the `other_exchange` object.  Every rank takes the same path (a
    # set-up failure is agreed on across ranks before anybody raises, dist.make_p2p_exchange).
    other = None
    if (world > 1 or args.force_dp) and not args.one_exchange and args.num_layers == 2 and args.hidden_dim == 64 and not args.no_fused_mlp:
        cur = D.allreduce_choice()
        alt = "p2p" if cur == "rccl" else "rccl"
        os.environ["AURPPO_DP_ALLREDUCE"] = alt
        other = {"exchange": alt}
        try:
            o_main = run_workload(args, rank, world, dev, args.envs_per_gpu, args.steps, args.warmup, False)
            other["collective_backend"] = getattr(o_main["agent"], "collective", None)
            other["main"] = {"ms_per_step": o_main["dt"] / args.steps * 1e3,
                             "value": world * o_main["N"] * args.num_steps * args.steps / o_main["dt"],
                             "update_launch": "hipGraph" if o_main["agent"]._graph is not None else "eager",
                             "graph_fallback": o_main["agent"].graph_fallback}
            if shard is not None:
                o_sh = run_workload(args, rank, world, dev, args.shard_envs_per_gpu, args.steps, args.warmup, False)
                other["config4_shard"] = {"ms_per_step": o_sh["dt"] / args.steps * 1e3,
                                          "value": world * o_sh["N"] * args.num_steps * args.steps / o_sh["dt"]}
            for tag, r_alt, r_pri in (("main", o_main, run), ("config4_shard", o_sh if shard is not None else None, shard)):
                if r_alt is None or r_pri is None or r_alt["gpu_first"] is None or r_pri["gpu_first"] is None:
                    continue
                wa, wp = r_alt["gpu_first"]["weights"], r_pri["gpu_first"]["weights"]
                err = max(float(np.abs(wa[k] - wp[k]).max()) for k in wa)
                other[tag]["first_update_max_abs_weight_diff_vs_primary"] = err
                other[tag]["agrees_with_primary"] = bool(err <= 2e-6)
                p2p = getattr((r_alt if alt == "p2p" else r_pri)["agent"], "_p2p", None)
                if p2p is not None:
                    other[tag]["p2p_status"] = p2p.status()
        except Exception as e:       # (the same exception on every rank, or none: see above)
            other["error"] = f"{type(e).__name__}: {e}"[:400]
        finally:
            os.environ["AURPPO_DP_ALLREDUCE"] = cur
    if rank != 0:
        D.shutdown()
        return
    agent, dt, N = run["agent"], run["dt"], run["N"]
    T, Dm, A = args.num_steps, args.obs_dim, args.act_dim
    probe, mlp_events, side, perm_events = run["probe"], run["mlp_events"], run["side"], run["perm_events"]
    env_steps = world * N * T * args.steps
    M = agent.minibatch_size

    traffic_src = {}

    def pmc(name, key=None):
        """PMC bytes per launch, collected (separate --pmc passes) on the default workload only (``key``: the net shape of
        a K7w run, for which only some shapes were collected).  Not measured in THIS run: ``traffic_source`` names the
        committed file and the commit it was last changed in; any other workload gets null."""
        if (M, Dm, A) != (131072, 64, 6):
            traffic_src["why_null"] = "PMC passes were collected at M=131072, D=64, A=6 only"
            return None
        path = os.path.join(ROOT, "profiles", name)
        try:
            d = json.load(open(path))
            val = (d[key] if key else d).get("hbm_bytes_per_launch")
        except Exception as e:
            traffic_src["why_null"] = f"profiles/{name}: {e}"
            return None
        traffic_src["file"] = f"profiles/{name}" + (f"[{key}]" if key else "")
        try:
            import subprocess
            traffic_src["git_sha"] = subprocess.run(["git", "-C", ROOT, "log", "-n", "1", "--format=%h", "--", path],
                                                    capture_output=True, text=True, timeout=10).stdout.strip() or None
        except Exception:
            traffic_src["git_sha"] = None
        if not traffic_src["git_sha"]:      # (the GPU box gets a snapshot without .git: the file names the commit it was collected at)
            traffic_src["git_sha"] = (d[key] if key else d).get("collected_at_commit") or d.get("collected_at_commit")
        traffic_src["how"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this bench (tools/collect_profiles.sh), FETCH_SIZE x 2 (gfx950)"
        return val

    B = N * T
    bytes_8d = (20 + args.epochs * (8 * Dm + 8 * A + 84)) * B            # 2 596 B/env-step at D=64, A=6, E=4
    ms_step = dt / args.steps * 1e3
    hbm_8d = {"bytes_per_step": bytes_8d, "achieved_GBs": round(bytes_8d / (ms_step * 1e-3) / 1e9, 1), "peak_GBs": HBM_PEAK_GBS,
              "frac": round(bytes_8d / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
              "note": "SURVEY 8d's algorithmic HBM bytes of GAE + E epochs of shuffle/gather/loss (policy nets excluded) over "
                      "the whole measured step.  north_star's 40 % of HBM presumes the nets run elsewhere; with the fp32 "
                      "actor/critic fused into the step (K7) the floor is 16 x 11.09 GFLOP / 157.3 TFLOP/s = 1.13 ms per "
                      "update, i.e. <= 15 % of this HBM figure is attainable at all: the step is MFMA-bound, see the main view"}
    side_stream = None
    if perm_events and len(perm_events) >= 3 and len(side) >= 3:
        # K2 for update u+1 is enqueued at the start of update u: busy = its own start -> end on the side stream (the
        # shortest period the shuffle pipeline could sustain); slack = how long before the main stream finished update u
        # the permutations of u+1 were ready
        k = min(len(perm_events), len(side))
        busy = [a.elapsed_time(b) for a, b in perm_events[:k]]
        slack = [pe[1].elapsed_time(me) for pe, me in zip(perm_events[:k], side[:k])]
        if os.environ.get("AURPPO_BENCH_STEPTIMES"):     # diagnostic: every update's end-to-end interval on the main stream
            log("main-stream interval per update (ms): " + " ".join(f"{side[i].elapsed_time(side[i + 1]):.3f}" for i in range(len(side) - 1)))
            log("shuffle busy per update (ms): " + " ".join(f"{x:.3f}" for x in busy))
        side_stream = {"k2_period_ms": round(float(np.median(busy)), 4), "slack_ms": round(float(np.median(slack)), 4),
                       "main_period_ms": round(ms_step, 4),
                       "how": "HIP events on the shuffle side stream (start/end of each update's E shuffles) and on the main "
                              "stream (end of each update) in this same timed region; medians"}
    from aur_ppo_amd import hip_ops as H
    k7_variant = H.k7_variant() if hasattr(H, "k7_variant") else 2
    roofline = None
    if mlp_events:
        ms = float(np.mean([b.elapsed_time(e) for b, e in mlp_events]))
        flops = H.mlp_step_flops(agent._mlp, M)
        ach = flops / (ms * 1e-3) / 1e12
        kname = (f"{H.k7w_kernel_name(args.hidden_dim, Dm, args.num_layers)} (K7w" if agent._mlp.get("wide") else ("k_mlp_step3" if k7_variant == 3 else "k_mlp_step2") + " (K7")
        roofline = {"bound": "mfma", "kernel": kname + ": gather + actor/critic forward + PPO loss + backward)",
                    "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                    "traffic": (pmc("mlp_wide_pmc.json", f"{args.num_layers}x{args.hidden_dim}") if agent._mlp.get("wide")
                                else pmc("mlp3_pmc.json" if k7_variant == 3 else "mlp_pmc.json")),
                    "traffic_source": traffic_src,
                    "flops_per_launch": flops, "avg_launch_us": round(ms * 1e3, 2), "launches_timed": len(mlp_events),
                    "algorithmic_hbm_bytes_per_launch": M * (4 * (Dm + A + 4) + 4),
                    "how": f"hipEvent pair recorded inside the library around the K7 kernel, one extra stand-alone "
                           f"launch every {run['every']}th step on the update's own minibatch"
                           + (" (the update itself is a hipGraph)" if agent._graph is not None else "")}
        wide_bf3 = bool(agent._mlp.get("wide")) and H.k7w_kernel(args.hidden_dim, Dm) == 3
        if wide_bf3:
            # k_mlpw3_step runs on the bf16 pipe too: priced the same way (6 x the algorithmic FLOPs against 2 500 TFLOP/s)
            roofline["vs_fp32_mfma"] = {"achieved": roofline["achieved"], "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": roofline["frac"],
                                        "what": "the algorithmic fp32 FLOPs priced at the fp32 MFMA peak the kernel does NOT run on"}
            ach6 = 6.0 * flops / (ms * 1e-3) / 1e12
            roofline.update(achieved=round(ach6, 1), peak=MFMA_BF16_PEAK_TFLOPS, frac=round(ach6 / MFMA_BF16_PEAK_TFLOPS, 4),
                            dtype_of_peak="bf16 MFMA dense (each fp32 product = 6 bf16 products, fp32 accumulate)")
        if k7_variant != 2 and not agent._mlp.get("wide"):
            # The kernel runs on the BF16 matrix pipe: six bf16 products per fp32 product.  The roof it is held to is that
            # pipe's (2.5 PFLOP/s dense): `achieved` = 6 x the algorithmic fp32 FLOPs / duration, `frac` against 2 500; the
            # same FLOPs priced at the fp32 MFMA peak (a speed-up statement, not a roofline: it can exceed 1) moves to
            # `vs_fp32_mfma`; `pipe` counts what the kernel actually ISSUES (padding included), the number the
            # SQ_VALU_MFMA_BUSY_CYCLES counter of profiles/r04/mlp3_mfma_pmc.json checks.
            issued = H.mlp_step_issued_bf16_flops(agent._mlp, M)
            roofline["vs_fp32_mfma"] = {"achieved": roofline["achieved"], "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": roofline["frac"],
                                        "what": "the algorithmic fp32 FLOPs priced at the fp32 MFMA peak the kernel does NOT run on"}
            ach6 = 6.0 * flops / (ms * 1e-3) / 1e12
            roofline.update(achieved=round(ach6, 1), peak=MFMA_BF16_PEAK_TFLOPS, frac=round(ach6 / MFMA_BF16_PEAK_TFLOPS, 4),
                            dtype_of_peak="bf16 MFMA dense (each fp32 product = 6 bf16 products, fp32 accumulate)")
            ach_i = issued / (ms * 1e-3) / 1e12
            roofline["pipe"] = {"issued_bf16_flops_per_launch": issued, "achieved": round(ach_i, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": round(ach_i / MFMA_BF16_PEAK_TFLOPS, 4),
                                "padding_over_6x_algorithmic": round(issued / (6.0 * flops), 4),
                                "what": "bf16 MFMA FLOPs the kernel issues per launch (6 x useful + head padded to 16 outputs; "
                                        "hip_ops.mlp_step_issued_bf16_flops) / duration / 2 500 TFLOP/s; counter check: "
                                        "profiles/r04/mlp3_mfma_pmc.json"}
            # the plain-fp32 number beside it: the same minibatch through k_mlp_step2 (v_mfma_f32_32x32x2_f32), stand-alone launches
            prev = os.environ.get("AURPPO_K7_VARIANT")
            os.environ["AURPPO_K7_VARIANT"] = "2"
            H.reload_knobs()
            try:
                evs = []
                for _ in range(6):
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    agent.probe_mlp_step(ev)
                    evs.append(ev)
                torch.cuda.synchronize()
                ms2 = float(np.mean([b.elapsed_time(e) for b, e in evs[1:]]))
                roofline["plain_f32_mfma"] = {"kernel": "k_mlp_step2 (v_mfma_f32_32x32x2_f32, fp32 operands)", "avg_launch_us": round(ms2 * 1e3, 2),
                                              "achieved": round(flops / (ms2 * 1e-3) / 1e12, 2), "frac": round(flops / (ms2 * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                              "how": "5 stand-alone launches on the same minibatch after the timed region (AURPPO_K7_VARIANT=2)"}
            finally:
                if prev is None:
                    os.environ.pop("AURPPO_K7_VARIANT", None)
                else:
                    os.environ["AURPPO_K7_VARIANT"] = prev
                H.reload_knobs()
            roofline["arithmetic"] = ("fp32 operands as three bf16 planes each, six v_mfma_f32_32x32x16_bf16 products per "
                                      "K = 16 (dropped terms <= 2^-24 relative), fp32 accumulate; `peak` is the bf16 pipe's")
    elif probe.pairs:
        gather_bytes = M * (8 * Dm + 8 * A + 36)          # idx + 6 streams read + written
        g_ms = probe.mean_ms()
        ach = gather_bytes / (g_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "k_gather", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc("gather_pmc.json"),
                    "bytes_per_launch": gather_bytes, "avg_launch_us": round(g_ms * 1e3, 2),
                    "launches_timed": len(probe.pairs),
                    "how": f"HIP-event pairs around one stand-alone launch every {run['every']}th step of the update's own gather"}
    if roofline is not None:
        roofline["hbm_8d"] = hbm_8d
        roofline["side_stream"] = side_stream

    def launch_of(ag):
        launch = "hipGraph" if ag._graph is not None else "eager"
        if ag.graph_fallback:
            launch += f" (capture failed: {ag.graph_fallback[:120]})"
        elif ag._graph is None and world > 1 and not D.collectives_capturable():
            launch += " (gloo rehearsal: a host-staged collective cannot be captured; RCCL runs captured)"
        return launch

    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    fused = agent._mlp is not None and not agent._mlp.get("wide")
    out = {"metric": "env-steps/sec through GAE+PPO-update at num_envs=4096,T=128; 1/2/4/8 GPU",
           "value": env_steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None,
           "dtype": ("f32 (3xbf16-split MFMA, fp32 accumulate)" if ((fused and k7_variant == 3) or
                                                                       (agent._mlp is not None and agent._mlp.get("wide")
                                                                        and H.k7w_kernel(args.hidden_dim, Dm) == 3)) else "f32"),
           "data": "synthetic",
           "config": {"workload": f"synthetic continuous obs_dim={Dm} act_dim={A}, num_envs={N}/GPU x {world} GPU, "
                                  f"T={T}, E={args.epochs}, {args.minibatches} minibatches/epoch (M={M}), "
                                  f"{args.num_layers}x{args.hidden_dim} tanh MLP actor+critic, Adam, random-init weights",
                      "global_num_envs": N * world, "num_steps": T,
                      "parallelism": f"env-shard dp{world}" + (" (one rank through the RCCL launch path)" if args.force_dp else ""),
                      "rccl_ranks": torch.distributed.get_world_size() if dist_on else 1,
                      "collective_backend": (getattr(agent, "collective", None) or (torch.distributed.get_backend() if dist_on else None)),
                      "process_group_backend": (torch.distributed.get_backend() if dist_on else None),
                      "update_launch": launch_of(agent),
                      "minibatch_step": ("K3 + torch nets + K5" if agent._mlp is None else
                                         "K7w fused MLP step + K6b" if agent._mlp.get("wide") else
                                         f"K7 fused MLP step (variant {k7_variant})")},
           "roofline": roofline,
           # a capture that raised leaves the update running eagerly on every rank: said HERE, not only inside config
           "graph_fallback": (agent.graph_fallback[:200] if agent.graph_fallback else None),
           "other_exchange": other}
    parity = None
    if world == 1 and not args.force_dp and args.cpu_baseline_updates > 0:
        out["cpu_baseline"], parity = cpu_baseline(args, run["data"], run["init_sd"], args.cpu_baseline_updates, run["gpu_first"])
    else:
        out["cpu_baseline"] = None
        if (world > 1 or args.force_dp) and not args.no_parity:
            parity = check_parity_sharded(args, run["hp"], N, world, run["init_sd"], run["gpu_first"], run["data"]["log_probs"])
    out["parity_checked"] = bool(parity["ok"]) if parity is not None else False
    out["parity"] = parity if parity is not None else "not run (--cpu-baseline-updates 0 / --no-parity)"
    shard_ok = True
    if shard is not None:
        sa, sN = shard["agent"], shard["N"]
        sp = None
        if not args.no_parity:
            if world == 1 and not args.force_dp:
                res, net = oracle_full_update(args, shard["hp"], sN, shard["data"], shard["init_sd"])
                sp = check_parity(shard["gpu_first"], res, net)
            else:
                sp = check_parity_sharded(args, shard["hp"], sN, world, shard["init_sd"], shard["gpu_first"],
                                          shard["data"]["log_probs"])
            shard_ok = bool(sp["ok"])
        out["config4_shard"] = {"what": "BASELINE config 4 / SURVEY 8e: the same step at 512 envs per GPU (the shard the 8-GPU "
                                        "metric is defined on), measured in this process right after the main workload",
                                "envs_per_gpu": sN, "n_gpus": world, "value": world * sN * T * args.steps / shard["dt"],
                                "unit": "env-steps/s", "ms_per_step": shard["dt"] / args.steps * 1e3, "steps": args.steps,
                                "minibatch": sa.minibatch_size, "update_launch": launch_of(sa),
                                "parity_checked": bool(sp["ok"]) if sp is not None else False, "parity": sp}
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    D.shutdown()
    if parity is not None and not parity["ok"]:
        sys.exit("bench.py: the GPU's first update does NOT match the oracle's -- see \"parity\" in the line above")
    if not shard_ok:
        sys.exit("bench.py: the 512-envs/GPU shard's first update does NOT match the oracle's -- see \"config4_shard\" above")


if __name__ == "__main__":
    main()
