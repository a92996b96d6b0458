"""``ppo`` -- the trainer of src/ppo.py with its API (``ppo(params).train()``, ``torch_buffer``,
``run_gae``, ``normal_advantage``, ``advantages``, ``rewards_to_go`` ...) and its numerics, with the
non-network part of the update on hand-written HIP kernels:

    rollout buffer (T,N,..) --K1 gae--> advantages/returns --K2 shuffle--> E x (B,) indices
      per minibatch: K3 fused gather -> policy.evaluate (PyTorch-ROCm) -> K4+K5 loss fwd+bwd
                     -> autograd through the nets -> [RCCL all-reduce] -> K6 clip -> Adam

What differs from the reference on purpose (MI355X-first, results unchanged):
  * nothing on the update path synchronises with the host: the per-minibatch ``.item()`` calls
    (src/ppo.py:234,246) become rows of a device-side scalar table read once per update;
  * all E epoch permutations of an update are generated up front on a side stream (they depend
    only on the RNG stream), overlapping the rollout / the previous update;
  * parameters and gradients live in one flat bucket (aur_ppo_amd/flat.py).
"""
from __future__ import annotations

import os
import random
import time

import numpy as np
import torch
from torch import nn

from . import dist as D
from .actor_critic import actor_critic
from .envs import make_vec_env
from .flat import FlatAdamMixin, FlatBucket
from .scalars import make_writer

device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


class torch_buffer:
    """Time-major rollout storage (src/ppo.py:20-39): ``states (T,N,*obs)``, ``actions (T,N,*act)``,
    ``log_probs / rewards / terminals / values (T,N)``; ``flatten`` returns zero-copy views with
    flat index ``t*N + n``."""

    def __init__(self, observation_shape, action_shape, num_steps, num_envs, device=device):
        self.observation_shape = tuple(observation_shape)
        self.action_shape = tuple(action_shape)
        z = lambda *s: torch.zeros(s, device=device)
        self.states = z(num_steps, num_envs, *self.observation_shape)
        self.actions = z(num_steps, num_envs, *self.action_shape)
        self.log_probs = z(num_steps, num_envs)
        self.rewards = z(num_steps, num_envs)
        self.terminals = z(num_steps, num_envs)
        self.values = z(num_steps, num_envs)

    def flatten(self, returns, advantages):
        b_obs = self.states.reshape((-1,) + self.observation_shape)
        b_logprobs = self.log_probs.reshape(-1)
        b_actions = self.actions.reshape((-1,) + self.action_shape)
        b_advantages = advantages.reshape(-1)
        b_returns = returns.reshape(-1)
        b_values = self.values.reshape(-1)
        return b_obs, b_logprobs, b_actions, b_advantages, b_returns, b_values


class ppo(FlatAdamMixin):
    """``ppo(params)`` as upstream (src/ppo.py:42-83).  Extra, optional ``params`` keys:
    ``obs_dim`` / ``act_dim`` / ``env_seed`` (Synthetic-* envs), ``log`` (False silences run logs),
    ``save`` (False skips the final pickle).  ``num_envs`` is the GLOBAL env count; under
    ``torch.distributed`` each rank keeps ``num_envs / world`` of them (aur_ppo_amd/dist.py).

    ``ops`` is a test seam: the module providing the kernels (default: ``aur_ppo_amd.hip_ops``,
    which raises without the built HIP library or without a GPU -- there is no fallback).
    Host-logic tests on a CPU box inject a stand-in built on the oracle; the product never does.
    """

    def __init__(self, params, ops=None, envs=None):
        self.params_dict = params
        self.all_steps = None
        self.minibatch_size = None
        for key, value in params.items():
            if key not in ("batch_size", "minibatch_size"):
                setattr(self, key, value)
        if ops is None:
            from . import hip_ops as ops   # fails loudly when the library is missing
        self.ops = ops
        self.device = torch.device(params.get("device", device))
        self.world = D.world_size()
        self.rank = D.rank()
        self.global_num_envs = int(self.num_envs)
        lo, hi = D.shard_envs(self.global_num_envs, self.rank, self.world)
        self.num_envs = hi - lo                      # local shard from here on
        self.env_lo = lo
        # derived sizes (src/ppo.py:61-64), per rank
        self.all_steps = self.num_steps * self.num_envs
        self.batch_size = int(self.num_envs * self.num_steps)
        self.minibatch_size = int(self.all_steps // self.num_minibatches)
        assert self.minibatch_size > 0, "num_minibatches exceeds the batch"
        self.num_updates = self.total_timesteps // (self.batch_size * self.world)
        self.run_name = f"{self.gym_id}__{self.exp_name}__{self.seed}__{int(time.time())}"
        p2 = dict(params)
        p2["rank"] = self.rank
        self.envs = envs if envs is not None else make_vec_env(self.gym_id, self.num_envs, self.continuous,
                                                               self.device, p2)
        self.state_dim = self.envs.single_observation_space.shape
        if self.continuous:
            self.action_dim = self.envs.single_action_space.shape
        else:
            self.action_dim = self.envs.single_action_space.n
        self.policy = actor_critic(self.state_dim[0], self.action_dim, self.hidden_dim, self.num_layers,
                                   self.dropout, self.continuous).to(self.device)
        if self.world > 1:                           # identical start on every rank
            for p in self.policy.parameters():
                torch.distributed.broadcast(p.data, src=0)
        self.buffer = torch_buffer(self.state_dim, self.envs.single_action_space.shape, self.num_steps,
                                   self.num_envs, self.device)
        self.bucket = FlatBucket(self.policy.parameters())
        # Adam(eps=1e-5) as upstream (src/ppo.py:80).  On the GPU it is built capturable with the learning
        # rate in a device scalar, so the whole update can be replayed as one hipGraph while the linear
        # anneal (src/ppo.py:195-198) still changes it between updates.
        if self.device.type == "cuda":
            self._lr_tensor = torch.tensor(float(self.learning_rate), device=self.device)
            self.optimizer = torch.optim.Adam(self.policy.parameters(), eps=1e-5, capturable=True, lr=self._lr_tensor)
        else:
            self._lr_tensor = None
            self.optimizer = torch.optim.Adam(self.policy.parameters(), lr=self.learning_rate, eps=1e-5)
        self._adam_setup()
        self.total_returns = []
        self.total_episode_lengths = []
        self.x_indices = []
        # device-side machinery of the update
        self.rng = None            # ops.MT19937, created by seed_all()
        self._perm_stream = None
        self._perm_ready = None
        self._perms = None
        self._perm_bufs = None
        self._perm_flip = 0
        self._last_perms = None
        n_steps = self.num_update_epochs * ((self.batch_size + self.minibatch_size - 1) // self.minibatch_size)
        self._scalars = torch.zeros((n_steps, ops.N_SCALARS), device=self.device)
        self._norms = torch.zeros(n_steps, device=self.device)
        self.last_update = None
        self._probe_outs = None
        self._probe_mlp_outs = None
        self._k2_flag = None
        # K7: fused gather + forward + loss + backward for the MLP actor-critic (None: per-op path)
        self._mlp = None
        if params.get("fused_mlp", True) and self.device.type == "cuda" and hasattr(ops, "mlp_layout"):
            self._mlp = ops.mlp_layout(self.policy, self.bucket)
        # the flat bucket holds the MLP policy and nothing else (up to alignment padding): K7 + clip + Adam can chain
        self._bucket_is_policy = self._mlp is not None and self.bucket.numel == self._mlp["n_params"]
        self._ro_state, self._ro_graph, self._ro_obs, self._ro_done, self._ro_out = 0, None, None, None, None   # captured rollout
        self._rec64 = None         # (B, 16) packed records for K7, rebuilt by _gae()
        self._graph = None         # captured update (hipGraph), see update()
        self._graph_state = 0      # 0: next update runs eagerly (warm-up), 1: capture, 2: replay
        self._perm_static = None
        # one process per GPU: the update is captured too when the collective is capturable (RCCL is; gloo stages
        # through the host and is not).  ``force_dp`` (test / rehearsal seam) takes the two-halves-around-the-all-reduce
        # path with a world of one, so a one-GPU box can rehearse the captured RCCL launch.
        self._dp = self.world > 1 or bool(params.get("force_dp", False))
        # AURPPO_DP_ALLREDUCE=p2p: the default 64-64 policy's gradient goes through the one-shot exchange over HIP-IPC peer memory
        # (csrc/p2p.hip) instead of the process group's all-reduce: nothing in the update then calls torch.distributed, so
        # it is capturable whatever the group's backend is
        self._p2p = None
        if (self._dp and D.allreduce_choice() == "p2p" and self._mlp is not None and not self._mlp.get("wide")
                and self._fused_adam and self._bucket_is_policy and hasattr(ops, "P2PExchange")):
            self._p2p = D.make_p2p_exchange(ops, self._mlp["n_params"], self.device)
        self._p2p_timeout = float(os.environ.get("AURPPO_P2P_TIMEOUT_S", "10"))      # a peer that never arrives raises p2p.status()
        self.collective = ("p2p (one-shot exchange over HIP-IPC peer memory, csrc/p2p.hip)" if self._p2p is not None
                           else (torch.distributed.get_backend() if self._dp and torch.distributed.is_initialized() else None))
        self.use_graph = (bool(params.get("hip_graph", True)) and self.device.type == "cuda"
                          and (not self._dp or self._p2p is not None or D.collectives_capturable()))
        self.graph_fallback = None  # why a captured update fell back to eager launches, if it did
        self._perm_events = None   # bench.py: [(start, end)] HIP events around each update's K2 work on the side stream
        self._rec = None           # (B,4) per-sample record written by K1
        self._rec_of = None
        self._probe = None         # bench.py hangs HIP-event pairs around the gather launches here
        self.first_grad_probe = None   # bench.py / tests: set to [] and the next update appends a copy of step 1's REDUCED gradient

    # ------------------------------------------------------------------ seeding / shuffle stream
    def seed_all(self, seed=1):
        """src/ppo.py:180-184 (seed hard-coded to 1 upstream, F3) + the device twin of numpy's stream."""
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        self.rng = self.ops.MT19937(seed, self.batch_size, self.device)
        self._perms = None

    def _prefetch_perms(self):
        """Enqueue the next update's E permutations (src/ppo.py:213-217) on a side stream, into one
        of two persistent (E, B) int32 buffers (allocated on the main stream, so no cross-stream
        allocator traffic).  The side stream first waits for everything already enqueued on the
        main stream -- which includes the last reader of the buffer being overwritten."""
        if self.rng is None:
            self.seed_all(1)
        if self.target_kl is not None:
            self._rng_snapshot = self.rng.get_state()      # early stop must rewind the stream
        if self._perm_bufs is None:
            self._perm_bufs = [torch.empty((self.num_update_epochs, self.batch_size), dtype=torch.int32,
                                           device=self.device) for _ in range(2)]
        out = self._perm_bufs[self._perm_flip]
        self._perm_flip ^= 1
        if self.device.type != "cuda":
            self._perms = self.rng.shuffle_epochs(self.batch_size, self.num_update_epochs, out=out)
            return
        if self._perm_stream is None:
            self._perm_stream = torch.cuda.Stream(device=self.device, priority=-1)   # latency-bound kernels: high priority
        self._perm_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._perm_stream):
            if self._perm_events is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(self._perm_stream)
            self._perms = self.rng.shuffle_epochs(self.batch_size, self.num_update_epochs, out=out)
            self._perm_ready = torch.cuda.Event(enable_timing=self._perm_events is not None)
            self._perm_ready.record(self._perm_stream)
            if self._perm_events is not None:
                self._perm_events.append((e0, self._perm_ready))

    def _take_perms(self):
        if self._perms is None:
            self._prefetch_perms()
        if self._perm_ready is not None:
            torch.cuda.current_stream(self.device).wait_event(self._perm_ready)
            self._perm_ready = None
        perms, self._perms = self._perms, None
        if self.target_kl is None:
            self._prefetch_perms()      # next update's shuffles overlap this update's compute
        return perms

    # ------------------------------------------------------------------ rollout (src/ppo.py:103-123)
    def make_env(self, gym_id, idx, capture_video):
        """Thunk that builds one wrapped gym environment (src/ppo.py:85-99): episode statistics always, video recording for
        env 0 when asked, and for continuous control the clip-action / normalise / clip observation and reward wrappers in
        upstream's order.  Needs ``gym`` (or ``gymnasium``), which this image does not ship: the thunk is returned either
        way, as upstream's is, and raises ImportError when CALLED without it (the synthetic device-resident envs of
        ``aur_ppo_amd.envs.make_vec_env`` are what ``train()`` uses then)."""
        def thunk():
            try:
                import gym
            except ImportError:
                try:
                    import gymnasium as gym
                except ImportError as e:
                    raise ImportError("ppo.make_env: neither gym nor gymnasium is importable; use aur_ppo_amd.envs.make_vec_env") from e
            env = gym.make(gym_id)
            env = gym.wrappers.RecordEpisodeStatistics(env)
            if capture_video and idx == 0:
                env = gym.wrappers.RecordVideo(env, f"videos/{self.run_name}")
            if self.continuous:
                env = gym.wrappers.ClipAction(env)
                env = gym.wrappers.NormalizeObservation(env)
                env = gym.wrappers.TransformObservation(env, lambda obs: np.clip(obs, -10, 10))
                env = gym.wrappers.NormalizeReward(env)
                env = gym.wrappers.TransformReward(env, lambda reward: np.clip(reward, -10, 10))
            return env
        return thunk

    def rewards_to_go(self, step, next_obs, global_step, writer):
        if self._mlp is not None and hasattr(self.ops, "mlp_act"):
            # K8: forward of both nets, sampling, log-prob and the three buffer row stores in one launch
            cont = self._mlp["continuous"]
            pre = getattr(self, "_rollout_noise", None)      # train() draws the whole rollout's noise in one call
            if pre is not None and step < pre.shape[0]:
                noise = pre[step]
            else:
                noise = (torch.randn((self.num_envs, self._mlp["A"]), device=self.device) if cont
                         else torch.rand(self.num_envs, device=self.device))
            action, _, _ = self.ops.mlp_act(next_obs.to(self.device).contiguous(), noise, self.bucket.flat_param, self._mlp,
                                            self.buffer.actions[step], self.buffer.log_probs[step], self.buffer.values[step])
            if not cont:
                action = action.long()
        else:
            with torch.no_grad():
                action, logprob, _, value = self.policy.evaluate(next_obs.to(self.device))
                self.buffer.values[step] = value.flatten()
            self.buffer.actions[step] = action
            self.buffer.log_probs[step] = logprob
        if getattr(self.envs, "device_native", False):
            next_obs, reward, next_done, _, info = self.envs.step(action)
            self.buffer.rewards[step] = reward.view(-1)
        else:
            next_obs, reward, done, _, info = self.envs.step(action.cpu().numpy())
            self.buffer.rewards[step] = torch.as_tensor(np.asarray(reward), dtype=torch.float32).view(-1)
            next_obs = torch.as_tensor(np.asarray(next_obs), dtype=torch.float32).to(self.device)
            next_done = torch.as_tensor(np.asarray(done), dtype=torch.float32).to(self.device)
        if "final_info" in info.keys():
            for item in info["final_info"]:
                if item is not None:
                    writer.add_scalar("charts/episodic_return", item["episode"]["r"], global_step)
                    writer.add_scalar("charts/episodic_length", item["episode"]["l"], global_step)
                    self.total_returns.append(item["episode"]["r"])
                    self.total_episode_lengths.append(item["episode"]["l"])
                    self.x_indices.append(global_step)
                    break
        return next_obs, next_done

    def _rollout_steps(self, next_obs, next_done, global_step, writer):
        """The T rollout steps of src/ppo.py:201-205."""
        if self._mlp is not None and hasattr(self.ops, "mlp_act"):
            # one generator call per rollout instead of one per step (the sampling noise K8 consumes)
            self._rollout_noise = (torch.randn((self.num_steps, self.num_envs, self._mlp["A"]), device=self.device)
                                   if self._mlp["continuous"] else torch.rand((self.num_steps, self.num_envs), device=self.device))
        for step in range(0, self.num_steps):
            global_step += 1 * self.num_envs * self.world
            self.buffer.states[step] = next_obs
            self.buffer.terminals[step] = next_done
            next_obs, next_done = self.rewards_to_go(step, next_obs, global_step, writer)
        self._rollout_noise = None      # a stand-alone rewards_to_go() call draws fresh noise, as upstream does
        return next_obs, next_done, global_step

    def _rollout(self, next_obs, next_done, global_step, writer):
        """One rollout.  With a device-resident env whose step() is a fixed sequence of device ops (``capturable``), the MLP
        policy (K8) and one process, the T steps -- ~10 launches each, host-bound -- are captured once as a hipGraph and
        replayed: rollout 1 runs eagerly, rollout 2 is captured, later ones replay."""
        env = self.envs
        ok = (self.use_graph and self.world == 1 and self._mlp is not None and hasattr(self.ops, "mlp_act")
              and getattr(env, "device_native", False) and getattr(env, "capturable", False) and self.device.type == "cuda")
        if not ok or self._ro_state == 0:
            self._ro_state = 1 if ok else 0
            return self._rollout_steps(next_obs, next_done, global_step, writer)
        if self._ro_obs is None:
            self._ro_obs, self._ro_done = torch.empty_like(next_obs), torch.empty_like(next_done)
        self._ro_obs.copy_(next_obs)
        self._ro_done.copy_(next_done)
        if self._ro_state == 1:
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            for gen in env.generators():
                g.register_generator_state(gen)
            with torch.cuda.graph(g):
                o, d, _ = self._rollout_steps(self._ro_obs, self._ro_done, 0, None)
                self._ro_out = (o, d)
            self._ro_graph = g
            self._ro_state = 2
        self._ro_graph.replay()
        return self._ro_out[0], self._ro_out[1], global_step + self.num_steps * self.num_envs * self.world

    # ------------------------------------------------------------------ advantages (src/ppo.py:125-166)
    def _gae(self, next_value, next_done, mode):
        """K1 + pack: besides (returns, advantages) the kernel leaves the per-sample record
        {old_logp, A, R, V} in ``self._rec`` for the packed gather / loss path of ``update``."""
        b = self.buffer
        if self._rec is None:
            self._rec = torch.empty((self.batch_size, 4), device=self.device)
        ret, adv = self.ops.gae(b.rewards, b.values, b.terminals, next_value.contiguous(), next_done.contiguous(),
                                self.gamma, self.gae_lambda, mode, log_probs=b.log_probs, rec=self._rec)
        self._rec_of = (ret, adv)
        if self._mlp is not None and hasattr(self.ops, "pack_records"):
            # K7 reads a sample's record and action row from one 64-B line (3 cache lines per sample instead of 4)
            acts = b.actions.reshape(self.batch_size, -1)
            if acts.shape[1] <= 12:
                self._rec64 = self.ops.pack_records(self._rec, acts, out=self._rec64)
        return ret, adv

    def run_gae(self, next_value, next_done):
        return self._gae(next_value, next_done, self.ops.GAE)

    def normal_advantage(self, next_value, next_done):
        return self._gae(next_value, next_done, self.ops.NORMAL_ADV)

    def advantages(self, next_obs, next_done):
        with torch.no_grad():
            if self._mlp is not None and hasattr(self.ops, "mlp_act"):
                _, _, next_value = self.ops.mlp_act(next_obs.contiguous(), None, self.bucket.flat_param, self._mlp)
            else:
                next_value = self.policy.value(next_obs)
            if self.gae:
                returns, advantages = self.run_gae(next_value, next_done)
            else:
                returns, advantages = self.normal_advantage(next_value, next_done)
        return returns, advantages

    # ------------------------------------------------------------------ update (src/ppo.py:210-273)
    def update(self, returns, advantages):
        """E epochs x minibatches of the clipped-surrogate step over the current buffer.  Returns
        the number of optimizer steps taken; per-step scalars are left in ``self._scalars``.

        On one GPU without ``target_kl`` the whole update (all gathers, forward/backward passes, loss
        kernels, clips and Adam steps -- ~1750 launches at the BASELINE size) is captured once into a
        hipGraph and replayed: update 1 runs eagerly (it also creates the optimizer state), update 2
        is captured, later updates replay.  The graph reads the permutations from a static buffer."""
        self._adopt_lr()
        packed = self._rec_of is not None and self._rec_of[0] is returns and self._rec_of[1] is advantages
        perms = self._take_perms()
        self._last_perms = perms        # complete and valid: what the probes below may index with
        graphable = (self.use_graph and packed and self.target_kl is None and self._probe is None
                     and self.first_grad_probe is None)
        if not graphable:
            return self._update_body(returns, advantages, perms, packed)
        if self._graph_state == 0:
            self._graph_state = 1
            return self._update_body(returns, advantages, perms, packed)
        if self._perm_static is None:
            self._perm_static = torch.empty_like(perms)
        self._perm_static.copy_(perms)
        if self._graph_state == 1:
            # the packed body reads only static storage: the rollout buffer, self._rec, self._perm_static
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            try:
                # with a collective inside, other threads of the process group (its watchdog) may touch the runtime
                # while this thread captures: keep the capture's error checking to this thread
                with torch.cuda.graph(g, capture_error_mode="thread_local" if self._dp else "global"):
                    self._graph_steps = self._update_body(returns, advantages, self._perm_static, packed)
            except Exception as e:                      # every rank captures the same calls, so every rank lands here
                if not self._dp:
                    raise
                self.graph_fallback = f"{type(e).__name__}: {e}"
                self.use_graph = False
                torch.cuda.synchronize(self.device)
                return self._update_body(returns, advantages, perms, packed)
            self._graph = g
            self._graph_state = 2
        self._graph.replay()
        return self._graph_steps

    def _update_body(self, returns, advantages, perms, packed):
        ops = self.ops
        b_obs, b_logprobs, b_actions, b_advantages, b_returns, b_values = self.buffer.flatten(returns, advantages)
        # packed path when (returns, advantages) are the tensors K1 just produced; otherwise (a caller
        # handing in its own) the six separate streams of buffer.flatten()
        srcs = [b_obs, b_actions, self._rec] if packed else [b_obs, b_actions, b_logprobs, b_advantages, b_returns, b_values]
        vmode = ops.VLOSS_CLIPPED if self.clip_vloss else ops.VLOSS_OLDVALUES   # src/ppo.py:250-261 (F8)
        B, M = self.batch_size, self.minibatch_size
        step = 0
        # single process, MLP policy, fused Adam over exactly the policy's bucket: K7 + clip + Adam chained, three
        # launches per minibatch (each call also prepares the statistics of the slice that follows it)
        chain = (packed and self._mlp is not None and not self._dp and self._fused_adam
                 and self._bucket_is_policy and hasattr(ops, "mlp_ppo_minibatch"))
        # one process per GPU: the same chain in two halves around the gradient all-reduce (SUM; the 1/W rides in the
        # apply kernel), three launches + one collective per minibatch
        chain_dp = (packed and self._mlp is not None and not self._mlp.get("wide") and self._dp and self._fused_adam
                    and self._bucket_is_policy and hasattr(ops, "mlp_ppo_grad"))
        k7_act, k7_rec = (None, self._rec64) if (packed and self._rec64 is not None) else (b_actions, self._rec)
        starts = list(range(0, B, M))
        for ep in range(self.num_update_epochs):
            idx_ep = perms[ep]
            for si, start in enumerate(starts):
                mb_inds = idx_ep[start:start + M]
                if chain or chain_dp:
                    if si + 1 < len(starts):
                        nxt = idx_ep[starts[si + 1]:starts[si + 1] + M]
                    elif ep + 1 < self.num_update_epochs:
                        nxt = perms[ep + 1][0:M]
                    else:
                        nxt = None
                    g = self.optimizer.param_groups[0]
                if chain_dp:
                    ops.mlp_ppo_grad(b_obs, k7_act, k7_rec, mb_inds, self.bucket.flat_param, self._mlp,
                                     self.bucket.flat_grad, self.clip_coeff, self.entropy_coeff, self.value_coeff,
                                     self.norm_adv, vmode, self._scalars[step], self._adam_t, chained=step > 0)
                    if self._p2p is not None:
                        # one launch, one hop: every rank reads its peers' published gradients and forms the mean in rank order
                        n = self._mlp["n_params"]
                        self._p2p.allreduce_mean_(self.bucket.flat_grad, n, self._adam_t, timeout_s=self._p2p_timeout)
                        if step == 0 and self.first_grad_probe is not None:
                            self.first_grad_probe.append(self.bucket.flat_grad.detach().clone())
                        ops.mlp_ppo_apply_parts(self.bucket.flat_param, self.bucket.flat_grad, self._adam_m, self._adam_v, self._mlp,
                                                self._lr_tensor, self._adam_t, self.max_grad_norm, g["betas"], g["eps"],
                                                self._norms[step:step + 1], self._p2p.parts(n), rec=k7_rec, next_idx=nxt)
                        step += 1
                        continue
                    D.allreduce_sum_(self.bucket.flat_grad, self.world, force=True)
                    if step == 0 and self.first_grad_probe is not None:
                        self.first_grad_probe.append(self.bucket.flat_grad.detach().clone() / self.world)
                    ops.mlp_ppo_apply(self.bucket.flat_param, self.bucket.flat_grad, self._adam_m, self._adam_v, self._mlp,
                                      self._lr_tensor, self._adam_t, self.max_grad_norm, g["betas"], g["eps"],
                                      self._norms[step:step + 1], grad_scale=1.0 / self.world, rec=k7_rec, next_idx=nxt)
                    step += 1
                    continue
                if chain:
                    ops.mlp_ppo_minibatch(b_obs, k7_act, k7_rec, mb_inds, self.bucket.flat_param, self._mlp,
                                          self.bucket.flat_grad, self.clip_coeff, self.entropy_coeff, self.value_coeff,
                                          self.norm_adv, vmode, self._scalars[step], self._adam_m, self._adam_v,
                                          self._lr_tensor, self._adam_t, self.max_grad_norm, g["betas"], g["eps"],
                                          self._norms[step:step + 1], next_idx=nxt, chained=step > 0)
                    step += 1
                    continue
                if packed and self._mlp is not None:
                    # one fused launch: rows are read through the permutation, gradients land in the bucket
                    ops.mlp_ppo_step(b_obs, k7_act, k7_rec, mb_inds, self.bucket.flat_param, self._mlp,
                                     self.bucket.flat_grad, self.clip_coeff, self.entropy_coeff, self.value_coeff,
                                     self.norm_adv, vmode, self._scalars[step])
                    D.allreduce_mean_(self.bucket.flat_grad, self.world, force=self._dp)
                    if step == 0 and self.first_grad_probe is not None:
                        self.first_grad_probe.append(self.bucket.flat_grad.detach().clone())
                    self._clip_and_step(self._norms[step:step + 1])
                    step += 1
                    continue
                mb = ops.gather(mb_inds, srcs, probe=self._probe) if self._probe is not None else ops.gather(mb_inds, srcs)
                _, newlogprob, entropy, newvalue = self.policy.evaluate(mb[0], mb[1])
                if packed:
                    loss = ops.ppo_loss_packed(newlogprob, newvalue, entropy, mb[2], self.clip_coeff,
                                               self.entropy_coeff, self.value_coeff, self.norm_adv, vmode,
                                               self._scalars[step])
                else:
                    loss = ops.ppo_loss(newlogprob, newvalue, entropy, mb[2], mb[3], mb[5], mb[4], self.clip_coeff,
                                        self.entropy_coeff, self.value_coeff, self.norm_adv, vmode,
                                        self._scalars[step])
                self.bucket.zero_grad()
                loss.backward()
                D.allreduce_mean_(self.bucket.flat_grad, self.world)
                self._clip_and_step(self._norms[step:step + 1])
                step += 1
            if self.target_kl is not None:
                # the reference compares the LAST minibatch's approx_kl (src/ppo.py:271-273)
                kl = self._scalars[step - 1, ops.S_KL].clone()
                if self.world > 1:
                    torch.distributed.all_reduce(kl)
                    kl /= self.world
                if float(kl) > self.target_kl:
                    # epochs after the break never shuffled upstream: rewind the RNG stream to here
                    self._rewind_rng(ep)
                    break
        return step

    def probe_gather(self, probe):
        """One stand-alone launch of exactly the update's first-minibatch gather (same index slice,
        same sources) with ``probe.begin()/end()`` around the C call.  bench.py uses it to time K3 with
        HIP events inside the timed region when the update itself runs as a hipGraph (events cannot be
        read back from inside a captured graph)."""
        perms = self._perm_static if self._perm_static is not None else self._last_perms
        srcs = [self.buffer.states.reshape((-1,) + self.buffer.observation_shape),
                self.buffer.actions.reshape((-1,) + self.buffer.action_shape), self._rec]
        if self._probe_outs is None:
            self._probe_outs = [torch.empty((self.minibatch_size,) + tuple(t.shape[1:]), device=self.device) for t in srcs]
        self.ops.gather(perms[0][:self.minibatch_size], srcs, self._probe_outs, probe=probe)

    def probe_mlp_step(self, events):
        """One stand-alone K7 launch on the update's own first minibatch (gradients go to a scratch
        bucket), with ``events`` recorded right around ``k_mlp_step`` inside the library call."""
        perms = self._perm_static if self._perm_static is not None else self._last_perms
        if self._probe_mlp_outs is None:
            self._probe_mlp_outs = [torch.empty_like(self.bucket.flat_grad), torch.empty(self.ops.N_SCALARS, device=self.device)]
        vmode = self.ops.VLOSS_CLIPPED if self.clip_vloss else self.ops.VLOSS_OLDVALUES
        acts, rec = (None, self._rec64) if self._rec64 is not None else (
            self.buffer.actions.reshape((-1,) + self.buffer.action_shape), self._rec)
        self.ops.mlp_ppo_step(self.buffer.states.reshape((-1,) + self.buffer.observation_shape), acts, rec,
                              perms[0][:self.minibatch_size], self.bucket.flat_param, self._mlp, self._probe_mlp_outs[0],
                              self.clip_coeff, self.entropy_coeff, self.value_coeff, self.norm_adv, vmode,
                              self._probe_mlp_outs[1], events=events)

    def _rewind_rng(self, last_epoch_run):
        """Early stop at epoch e: upstream has drawn e+1 shuffles this update, we pre-drew E.
        Re-create the stream position by replaying from the snapshot taken at update start."""
        key, pos = self._rng_snapshot
        self.rng.set_state(key, pos)
        self.rng.shuffle_epochs(self.batch_size, last_epoch_run + 1)
        self._perms = None

    # ------------------------------------------------------------------ train (src/ppo.py:169-300)
    def train(self):
        log = self.params_dict.get("log", True)
        if getattr(self, "track", False):
            import wandb
            wandb.init(project="ppo", sync_tensorboard=True, config=None, name=self.run_name, save_code=True)
        writer = make_writer(f"runs/{self.run_name}", write=log and self.rank == 0)
        self.writer = writer
        writer.add_text("hyperparameters", "|param|value|\n|-|-|\n%s" % (
            "\n".join([f"|{key}|{str(self.params_dict[key])}|" for key in self.params_dict])))
        self.seed_all(1)
        global_step = 0
        start_time = time.time()
        next_obs = self.envs.reset(seed=list(range(self.env_lo, self.env_lo + self.num_envs)))[0]
        next_obs = torch.as_tensor(next_obs, dtype=torch.float32).to(self.device)
        next_done = torch.zeros(self.num_envs, device=self.device)
        first_update = 1
        if self.params_dict.get("resume"):          # not upstream: continue an interrupted run (weights, Adam, RNG, update)
            first_update = self.load_checkpoint(self.params_dict["resume"]) + 1
            global_step = (first_update - 1) * self.batch_size * self.world
        ck_path, ck_every = self.params_dict.get("checkpoint_path"), int(self.params_dict.get("checkpoint_every", 0))
        for update in range(first_update, self.num_updates + 1):
            if self.anneal_lr:
                frac = 1.0 - (update - 1.0) / self.num_updates
                self.set_lr(frac * self.learning_rate)
            if self._perms is None:
                self._prefetch_perms()              # overlaps the rollout below
            next_obs, next_done, global_step = self._rollout(next_obs, next_done, global_step, writer)
            returns, advantages = self.advantages(next_obs, next_done)
            n_steps = self.update(returns, advantages)
            self._log_update(writer, returns, n_steps, global_step, start_time)
            if ck_path and ck_every > 0 and update % ck_every == 0 and self.rank == 0:
                self.save_checkpoint(ck_path, update=update)
        self.envs.close()
        writer.close()
        if self.params_dict.get("save", True) and self.rank == 0:
            torch.save(self.policy, "actor_critic_" + str(self.num_layers) + ".pt")
        if len(self.total_returns) >= 10 and self.rank == 0 and log:
            self.plot_episodic_returns(np.array(self.total_returns), np.array(self.x_indices), "episodic returns")
            self.plot_episodic_returns(np.array(self.total_episode_lengths), np.array(self.x_indices), "episodic lengths")
        return self.total_returns, self.total_episode_lengths, self.x_indices

    def _log_update(self, writer, returns, n_steps, global_step, start_time):
        """One host read per update: the scalar table, then the reference's tags (src/ppo.py:277-292)."""
        ops = self.ops
        b_values, b_returns = self.buffer.values.reshape(-1), returns.reshape(-1)
        var_y = b_returns.var(unbiased=False)
        ev = 1 - (b_returns - b_values).var(unbiased=False) / var_y
        if self._k2_flag is None:
            self._k2_flag = torch.zeros(1, device=self.device)
        if self.rng is not None and hasattr(self.rng, "status_into"):
            self.rng.status_into(self._k2_flag)       # sticky "a shuffle ran out of draws": rides in the same read
        table = torch.cat([self._scalars[:n_steps].reshape(-1), self._norms[:n_steps], var_y.view(1), ev.view(1),
                           self._k2_flag]).cpu()
        if float(table[-1]) != 0.0:
            raise RuntimeError("K2: a shuffle consumed more draws than were pre-generated (beyond 12 sigma of numpy's "
                               "rejection sampling); the minibatch permutations of this update are invalid")
        table = table[:-1]
        sc = table[:n_steps * ops.N_SCALARS].view(n_steps, ops.N_SCALARS).numpy()
        var_y, ev = float(table[-2]), float(table[-1])
        last = sc[-1]                                # logged values are the last minibatch's
        self.last_update = dict(scalars=sc, grad_norms=table[n_steps * ops.N_SCALARS:-2].numpy(),
                                explained_variance=(np.nan if var_y == 0 else ev))
        writer.add_scalar("charts/learning_rate", self.get_lr(), global_step)
        writer.add_scalar("losses/value_loss", last[ops.S_VL], global_step)
        writer.add_scalar("losses/policy_loss", last[ops.S_PG], global_step)
        writer.add_scalar("losses/entropy", last[ops.S_ENT], global_step)
        writer.add_scalar("losses/old_approx_kl", last[ops.S_OLD_KL], global_step)
        writer.add_scalar("losses/approx_kl", last[ops.S_KL], global_step)
        writer.add_scalar("losses/clipfrac", float(np.mean(sc[:, ops.S_CLIPFRAC])), global_step)
        writer.add_scalar("losses/explained_variance", self.last_update["explained_variance"], global_step)
        writer.add_scalar("charts/SPS", int(global_step / (time.time() - start_time)), global_step)

    # ------------------------------------------------------------------ plotting (src/ppo.py:303-321)
    def plot(self, loss, x_indices):
        import matplotlib.pyplot as plt
        plt.plot(np.array(x_indices), loss)
        plt.xlabel("Timestep")
        plt.ylabel("Total returns")
        plt.title("Episode Length over time")
        plt.show()

    def moving_average(self, data, window_size):
        return np.convolve(data, np.ones(window_size) / window_size, mode="valid")

    def plot_episodic_returns(self, episodic_returns, x_indices, title, window_size=10):
        import os
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        smoothed = self.moving_average(episodic_returns, window_size)
        plt.figure()
        plt.plot(x_indices, episodic_returns, label="Episodic Returns")
        plt.plot(x_indices[window_size - 1:], smoothed, label=f"Moving Average (Window Size = {window_size})", color="red")
        plt.title("Episodic Returns with Moving Average for " + self.gym_id)
        plt.xlabel("Timestep")
        plt.ylabel("Return")
        plt.legend()
        out_dir = "../plots" if os.path.isdir("../plots") else "plots"
        os.makedirs(out_dir, exist_ok=True)
        plt.savefig(f"{out_dir}/{title}_num_layers_{self.num_layers}_dropout_{self.dropout}_num_envs_"
                    f"{self.global_num_envs}_num_mb_{self.num_minibatches}.png")
        plt.close()
