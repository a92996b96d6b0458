"""``robot_ppo`` -- the BulletArm fork of the trainer (src/robot_ppo.py) with its API: image
``torch_buffer`` with ``true_actions``, ``store_returns``, ``rewards_to_go``, ``expert_rollout``,
``pretrain`` / ``pretrain_update``, ``run_gae(next_value, next_done, buffer, num_steps)``,
``normal_advantage(...)``, ``advantages(next_state, next_obs, next_done, buffer, num_steps)``,
``test_env``, ``update(buffer, update_epochs, batch_size, minibatch_size, policy_losses)``, ``train``.

Same HIP kernels as ``ppo`` (K1-K6).  Upstream behaviour kept, with the SURVEY finding it stems from:
  * ``run_gae`` never visits t = T-1 (src/robot_ppo.py:230, F4) -> K1 mode AURPPO_GAE_SKIP_LAST
    (params['fix_gae_bootstrap']=True selects the canonical recurrence instead);
  * un-clipped value loss regresses to the RETURNS (src/robot_ppo.py:390) -> AURPPO_VLOSS_RETURNS;
  * the reported value loss is already multiplied by value_coeff (src/robot_ppo.py:392);
  * ``clip_grad_norm_`` covers the ACTOR's parameters only (src/robot_ppo.py:401) -> K6 on the actor's
    slice of the flat gradient bucket (the actor is placed first in the bucket for that reason);
  * ``update`` slices minibatches with ``self.minibatch_size`` (src/robot_ppo.py:341);
  * the expert MSE term is between two buffer tensors (src/robot_ppo.py:397, F7): it has no gradient
    path and only shifted an un-returned ``loss`` value, so it is not computed;
  * ``pretrain_update`` likewise back-propagates into a buffer tensor, not the policy (F7): no parameter receives a
    gradient, upstream's ``optimizer.step()`` is a no-op there, and here the optimizer is not stepped at all.
Deliberately different: ``log_probs`` is stored as (T, N) -- upstream's (T, N, action_shape) buffer
only ran when num_envs == 5 by an accidental broadcast and then indexed the wrong elements (F5).
"""
from __future__ import annotations

import random
import time

import numpy as np
import torch
from torch import nn

from . import dist as D
from .envs import make_arm_envs
from .flat import FlatAdamMixin, FlatBucket
from .robot_actor_critic import robot_actor_critic
from .scalars import make_writer

device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


class torch_buffer:
    """src/robot_ppo.py:21-66: ``states (T,N)`` (gripper flag), ``observations (T,N,*obs)``,
    ``actions / true_actions (T,N,A)``, ``log_probs / rewards / terminals / values (T,N)``."""

    def __init__(self, state_shape, observation_shape, action_shape, num_steps, num_envs):
        self.state_shape = state_shape
        self.observation_shape = tuple(observation_shape)
        self.action_shape = action_shape
        self.states = torch.zeros((num_steps, num_envs))
        self.observations = torch.zeros((num_steps, num_envs) + self.observation_shape)
        self.actions = torch.zeros((num_steps, num_envs, action_shape))
        self.true_actions = torch.zeros((num_steps, num_envs, action_shape))
        self.log_probs = torch.zeros((num_steps, num_envs))
        self.rewards = torch.zeros((num_steps, num_envs))
        self.terminals = torch.zeros((num_steps, num_envs))
        self.values = torch.zeros((num_steps, num_envs))

    _FIELDS = ("states", "observations", "actions", "log_probs", "rewards", "terminals", "values", "true_actions")

    def load_to_device(self, dev=None):
        for f in self._FIELDS:
            setattr(self, f, getattr(self, f).to(dev if dev is not None else device))

    def load_to_cpu(self):
        for f in self._FIELDS:
            setattr(self, f, getattr(self, f).to("cpu"))

    def flatten(self, returns, advantages):
        T, N = self.states.shape
        return (self.states.view(T * N), self.observations.view((T * N,) + self.observation_shape),
                self.log_probs.reshape(-1), self.actions.view(T * N, -1), advantages.reshape(-1), returns.reshape(-1),
                self.values.reshape(-1), self.true_actions.view(T * N, -1))


class store_returns:
    """Per-env reward lists for the discounted-return chart (src/robot_ppo.py:69-83)."""

    def __init__(self, num_envs, gamma):
        self.gamma = gamma
        self.env_returns = [[] for _ in range(num_envs)]

    def add_value(self, i, reward):
        self.env_returns[i].append(reward)

    def calc_discounted_return(self, i):
        n = len(self.env_returns[i])
        R = 0
        for r in reversed(self.env_returns[i]):
            R = r + self.gamma * R
        self.env_returns[i] = []
        return R, n


class robot_ppo(FlatAdamMixin):
    def __init__(self, params, ops=None, envs=None, eval_envs=None):
        self.params_dict = params
        self.all_steps = None
        self.minibatch_size = None
        for key, value in params.items():
            if key not in ("batch_size", "minibatch_size"):
                setattr(self, key, value)
        for key, dflt in (("render", False), ("save_file_path", None), ("do_pretraining", True), ("anneal_exp", False),
                          ("expert_weight", 0.9), ("equivariant", False), ("pretrain_steps", 1000),
                          ("pretrain_batch_size", 8), ("track", False)):
            if not hasattr(self, key):
                setattr(self, key, dflt)
        if ops is None:
            from . import hip_ops as ops
        self.ops = ops
        self.device = torch.device(params.get("device", device))
        self.world, self.rank = D.world_size(), D.rank()
        self.global_num_envs = int(self.num_envs)
        lo, hi = D.shard_envs(self.global_num_envs, self.rank, self.world)
        self.num_envs, self.env_lo = hi - lo, lo
        self.all_steps = self.num_steps * self.num_envs
        self.batch_size = int(self.num_envs * self.num_steps)
        self.minibatch_size = int(self.all_steps // self.num_minibatches)
        assert self.minibatch_size != 0
        self.num_updates = self.total_timesteps // (self.batch_size * self.world)
        self.run_name = f"{self.gym_id}__{self.exp_name}__{self.seed}__{int(time.time())}"
        self.total_pretrain_steps = self.pretrain_steps * self.num_envs
        self.pretrain_minibatch_size = int(self.total_pretrain_steps // self.pretrain_batch_size)
        self.envs = envs if envs is not None else make_arm_envs(self.gym_id, self.num_envs, self.device, params, self.rank)
        self.eval_envs = eval_envs if eval_envs is not None else make_arm_envs(self.gym_id, 5, self.device, params, 1000)
        self.plot_index = 0
        # extra, optional params: obs_size / obs_channels -- (1, 128, 128) upstream; (3, 84, 84) is BASELINE config 5's shape
        self.obs_shape = (int(params.get("obs_channels", 1)), int(params.get("obs_size", 128)), int(params.get("obs_size", 128)))
        kw = dict(obs_shape=self.obs_shape)
        if "equiv_hidden" in params:           # extra: width of the build-defined equivariant nets (128 regular fields upstream)
            kw["n_hidden"] = int(params["equiv_hidden"])
        self.policy = robot_actor_critic(self.device, self.equivariant, **kw).to(self.device)
        self.expert = robot_actor_critic(self.device, self.equivariant, **kw).to(self.device)
        if self.device.type == "cuda" and bool(params.get("channels_last", False)):
            self.policy.memory_format = torch.channels_last      # NHWC activations from the first convolution on
        if self.world > 1:
            for p in self.policy.parameters():
                torch.distributed.broadcast(p.data, src=0)
        self.action_dim = 5
        self.state_dim = 1
        self.buffer = torch_buffer(self.state_dim, self.obs_shape, self.action_dim, self.num_steps, self.num_envs)
        self.buffer.load_to_device(self.device)
        self.pretrain_buffer = None        # allocated by pretrain() (upstream builds it eagerly, on the host)
        # actor first in the flat bucket: upstream clips the actor's gradient only (src/robot_ppo.py:401)
        actor_params = list(self.policy.actor.parameters())
        actor_ids = {id(p) for p in actor_params}
        rest = [p for p in self.policy.parameters() if id(p) not in actor_ids]
        self.bucket = FlatBucket(actor_params + rest)
        self.n_actor = sum(p.numel() for p in actor_params)
        if self.device.type == "cuda":
            self._lr_tensor = torch.tensor(float(self.learning_rate), device=self.device)
            self.optimizer = torch.optim.Adam(self.policy.parameters(), eps=1e-5, capturable=True, lr=self._lr_tensor)
        else:
            self._lr_tensor = None
            self.optimizer = torch.optim.Adam(self.policy.parameters(), lr=self.learning_rate, eps=1e-5)
        self._adam_setup()
        self.pretrain_optimizer = torch.optim.Adam(self.expert.actor.parameters(), lr=self.learning_rate, eps=1e-5)
        self.total_returns, self.total_episode_lengths, self.x_indices = [], [], []
        self.episodic_returns = store_returns(self.num_envs, self.gamma)
        self.rng = None
        self._rec = None
        n_steps = self.num_update_epochs * ((self.batch_size + self.minibatch_size - 1) // self.minibatch_size)
        self._scalars = torch.zeros((n_steps, ops.N_SCALARS), device=self.device)
        self._norms = torch.zeros(n_steps, device=self.device)

    # ------------------------------------------------------------------ helpers (Adam: flat.FlatAdamMixin)
    def seed_all(self, seed=1):
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        self.rng = self.ops.MT19937(seed, max(self.batch_size, self.total_pretrain_steps), self.device)

    # ---- checkpoint layout: upstream's three keys (src/robot_ppo.py:502-507) + what it forgets (``actor_logstd`` is a
    # parameter of robot_actor_critic itself, in neither sub-module's state dict) + the resume state
    def _checkpoint_nets(self):
        return [("actor_state", self.policy.actor), ("critic_state", self.policy.critic)]

    def _checkpoint_extra(self):
        ls = getattr(self.policy, "actor_logstd", None)       # the equivariant actor produces its own log-std
        return {"actor_logstd": ls.detach().cpu().clone()} if ls is not None else {}

    def _checkpoint_restore(self, sd):
        if "actor_logstd" in sd and hasattr(self.policy, "actor_logstd"):
            self.policy.actor_logstd.copy_(sd["actor_logstd"])

    # ------------------------------------------------------------------ rollout (src/robot_ppo.py:161-197)
    def rewards_to_go(self, step, next_state, next_obs, global_step, writer):
        with torch.no_grad():
            actions, unscaled, logprob, _, value = self.policy.evaluate(next_state.to(self.device), next_obs.to(self.device))
            self.buffer.values[step] = value.flatten()
        self.buffer.actions[step] = unscaled
        self.buffer.log_probs[step] = logprob
        with torch.no_grad():
            true_action, _scaled = self.policy.getActionFromPlan(self.envs.getNextAction().to(self.device))
        self.buffer.true_actions[step] = true_action
        next_states, next_obs, reward, done = self.envs.step(actions)
        self.buffer.rewards[step] = reward.view(-1)
        if not getattr(self.envs, "device_native", False) or bool(self.params_dict.get("episode_stats", False)):
            for i, rew in enumerate(reward.tolist()):
                self.episodic_returns.add_value(i, rew)
            for i, d in enumerate(done.tolist()):
                if d:
                    discounted_return, episode_length = self.episodic_returns.calc_discounted_return(i)
                    writer.add_scalar("charts/discounted_episodic_return", discounted_return, global_step)
                    writer.add_scalar("charts/episodic_length", episode_length, global_step)
                    break
        return next_states.to(self.device), next_obs.to(self.device), done.to(self.device)

    def expert_rollout(self, step, state, obs):
        pb = self.pretrain_buffer
        with torch.no_grad():
            unscaled, scaled = self.policy.getActionFromPlan(self.envs.getNextAction().to(self.device))
            _sa, unscaled_agent, logprob, _, value = self.policy.evaluate(state.to(self.device), obs.to(self.device))
            pb.actions[step] = unscaled_agent
            pb.values[step] = value.detach().flatten()
        pb.true_actions[step] = unscaled
        pb.log_probs[step] = logprob
        next_states, next_obs, reward, done = self.envs.step(scaled, auto_reset=True)
        pb.rewards[step] = reward.view(-1)
        return next_states, next_obs, done

    # ------------------------------------------------------------------ advantages (src/robot_ppo.py:224-271)
    def _gae(self, next_value, next_done, buffer, num_steps, mode):
        T = int(num_steps)
        rec = None
        if buffer is self.buffer:
            if self._rec is None:
                self._rec = torch.empty((self.batch_size, 4), device=self.device)
            rec = self._rec
        ret, adv = self.ops.gae(buffer.rewards[:T], buffer.values[:T], buffer.terminals[:T],
                                next_value.contiguous().to(self.device), next_done.contiguous().to(self.device),
                                self.gamma, self.gae_lambda, mode,
                                log_probs=buffer.log_probs[:T] if rec is not None else None, rec=rec)
        return ret, adv

    def run_gae(self, next_value, next_done, buffer, num_steps):
        mode = self.ops.GAE if self.params_dict.get("fix_gae_bootstrap", False) else self.ops.GAE_SKIP_LAST
        return self._gae(next_value, next_done, buffer, num_steps, mode)

    def normal_advantage(self, next_value, next_done, buffer, num_steps):
        return self._gae(next_value, next_done, buffer, num_steps, self.ops.NORMAL_ADV)

    def advantages(self, next_state, next_obs, next_done, buffer, num_steps):
        with torch.no_grad():
            next_value = self.policy.value(next_state.to(self.device), next_obs.to(self.device)).flatten()
            if self.gae:
                return self.run_gae(next_value, next_done, buffer, num_steps)
            return self.normal_advantage(next_value, next_done, buffer, num_steps)

    # ------------------------------------------------------------------ behavioural-cloning warm-up (:273-307)
    def pretrain(self):
        if self.pretrain_buffer is None:
            self.pretrain_buffer = torch_buffer(self.state_dim, self.obs_shape, self.action_dim, self.pretrain_steps,
                                                self.num_envs)
            self.pretrain_buffer.load_to_device(self.device)
        state, obs = self.envs.reset()
        done = torch.zeros(self.num_envs, device=self.device)
        for step in range(0, self.pretrain_steps):
            self.pretrain_buffer.states[step] = state
            self.pretrain_buffer.observations[step] = obs
            self.pretrain_buffer.terminals[step] = done
            state, obs, done = self.expert_rollout(step, state, obs)
        return state.to(self.device), obs.to(self.device), done.to(self.device)

    def pretrain_update(self, buffer, update_epochs, batch_size, minibatch_size):
        """As upstream: the MSE is taken between two buffer tensors, so its gradient lands on the buffer
        copy and no policy parameter moves (F7).  The shuffle draws still advance the RNG stream."""
        (_s, _o, _lp, b_actions, _a, _r, _v, b_true_actions) = buffer
        idx = self.ops.arange_i32(batch_size, self.device) if hasattr(self.ops, "arange_i32") else \
            torch.arange(batch_size, dtype=torch.int32, device=self.device)
        for _ep in range(update_epochs):
            self.rng.shuffle_(idx)
            for index in range(0, batch_size, minibatch_size):
                mb = idx[index:index + self.minibatch_size].long()
                expert_loss = nn.functional.mse_loss(b_actions[mb].requires_grad_(True), b_true_actions[mb])
                self.bucket.zero_grad()
                expert_loss.backward()
                # upstream calls optimizer.step() here (src/robot_ppo.py:321-325) with every policy gradient None
                # (zero_grad sets them to None in torch 2.1): the step skips every parameter, creates no Adam state
                # and counts nothing.  Stepping the flat bucket on all-zero gradients would advance Adam's step
                # counter, so the optimizer is left alone.

    def test_env(self, writer):
        test_returns = store_returns(self.num_envs, self.gamma)
        with torch.no_grad():
            state, obs = self.envs.reset()
            for step in range(100):
                scaled_agent, _u, _lp, _, _v = self.policy.evaluate(state.to(self.device), obs.to(self.device))
                state, obs, reward, done = self.envs.step(scaled_agent, auto_reset=True)
                for i, rew in enumerate(reward.tolist()):
                    test_returns.add_value(i, rew)
                for i, d in enumerate(done.tolist()):
                    if d:
                        discounted_return, episode_length = test_returns.calc_discounted_return(i)
                        writer.add_scalar("charts/test_discounted_episodic_return", discounted_return, step)
                        writer.add_scalar("charts/test_episodic_length", episode_length, step)

    # ------------------------------------------------------------------ update (src/robot_ppo.py:329-408)
    def update(self, buffer, update_epochs, batch_size, minibatch_size, policy_losses):
        assert minibatch_size != 0
        ops = self.ops
        (b_states, b_obs, b_logprobs, b_actions, b_advantages, b_returns, b_values, _b_true_actions) = buffer
        packed = (self._rec is not None and b_logprobs.data_ptr() == self.buffer.log_probs.data_ptr()
                  and batch_size == self.batch_size)
        if packed:
            srcs = [b_obs, b_actions, self._rec, b_states]
        else:
            srcs = [b_obs, b_actions, b_states, b_logprobs, b_advantages, b_returns, b_values]
        if self.rng is None:
            self.seed_all(1)
        # all epochs' permutations are drawn up front; upstream draws one per epoch it actually runs (src/robot_ppo.py:338), so an
        # early stop on target_kl must leave the generator where numpy's would be: snapshot here, replay on the break below
        snapshot = self.rng.get_state() if self.target_kl is not None else None
        perms = self.rng.shuffle_epochs(batch_size, update_epochs)
        self._adopt_lr()
        vmode = ops.VLOSS_CLIPPED if self.clip_vloss else ops.VLOSS_RETURNS     # src/robot_ppo.py:379-390
        M = self.minibatch_size                                                  # sic: not the argument (:341)
        step = 0
        for ep in range(update_epochs):
            for index in range(0, batch_size, minibatch_size):
                mb_inds = perms[ep][index:index + M]
                mb = ops.gather(mb_inds, srcs)
                if packed:
                    obs, act, rec, st = mb
                else:
                    obs, act, st, old_lp, adv, ret, val = mb
                _, _, newlogprob, entropy, newvalue = self.policy.evaluate(st, obs, act)
                if packed:
                    loss = ops.ppo_loss_packed(newlogprob, newvalue, entropy, rec, self.clip_coeff, self.entropy_coeff,
                                               self.value_coeff, self.norm_adv, vmode, self._scalars[step])
                else:
                    loss = ops.ppo_loss(newlogprob, newvalue, entropy, old_lp, adv, val, ret, self.clip_coeff,
                                        self.entropy_coeff, self.value_coeff, self.norm_adv, vmode, self._scalars[step])
                self.bucket.zero_grad()
                loss.backward()
                D.allreduce_mean_(self.bucket.flat_grad, self.world)
                self._clip_and_step(self._norms[step:step + 1], self.n_actor)
                step += 1
            if self.target_kl is not None:
                # the reference compares the LAST minibatch's approx_kl (src/robot_ppo.py:406-408); one process per GPU:
                # the mean over ranks, so that every rank leaves the epoch loop together (the collectives stay matched)
                kl = self._scalars[step - 1, ops.S_KL].clone()
                if self.world > 1:
                    torch.distributed.all_reduce(kl)
                    kl /= self.world
                if float(kl) > self.target_kl:
                    if ep + 1 < update_epochs:
                        self.rng.set_state(*snapshot)
                        self.rng.shuffle_epochs(batch_size, ep + 1)
                    break
        flag = torch.zeros(1, device=self.device)
        if hasattr(self.rng, "status_into"):
            self.rng.status_into(flag)                # sticky "a shuffle ran out of draws": rides in the same read
        both = torch.cat([self._scalars[:step].reshape(-1), flag]).cpu()
        if float(both[-1]) != 0.0:
            raise RuntimeError("K2: a shuffle consumed more draws than were pre-generated; this update's permutations are invalid")
        sc = both[:-1].view(step, ops.N_SCALARS)
        self._last_scalars = sc.numpy()
        last = sc[-1]
        policy_losses.extend(sc[:, ops.S_PG].tolist())
        clip_fracs = sc[:, ops.S_CLIPFRAC].tolist()
        # value_loss is reported already weighted (src/robot_ppo.py:392)
        return (last[ops.S_PG], last[ops.S_VL] * self.value_coeff, last[ops.S_ENT], last[ops.S_OLD_KL], last[ops.S_KL],
                clip_fracs)

    # ------------------------------------------------------------------ train (src/robot_ppo.py:412-511)
    def train(self):
        log = self.params_dict.get("log", True)
        writer = make_writer(f"runs/{self.gym_id}", write=log and self.rank == 0)
        self.writer = writer
        writer.add_text("hyperparameters", "|param|value|\n|-|-|\n%s" % (
            "\n".join([f"|{key}|{str(self.params_dict[key])}|" for key in self.params_dict])))
        self.seed_all(1)
        resume = self.params_dict.get("resume")
        if self.do_pretraining and not resume:      # a resumed run already holds the pre-trained (and further trained) weights
            self.policy.train()
            next_state, next_obs, next_done = self.pretrain()
            returns, advantages = self.advantages(next_state, next_obs, next_done, self.pretrain_buffer, self.pretrain_steps)
            flat = self.pretrain_buffer.flatten(returns, advantages)
            self.pretrain_update(flat, self.num_update_epochs, self.pretrain_batch_size, self.pretrain_minibatch_size)
            self.test_env(writer)
        global_step = 0
        start_time = time.time()
        next_state, next_obs = self.envs.reset()
        next_state, next_obs = next_state.to(self.device), next_obs.to(self.device)
        next_done = torch.zeros(self.num_envs, device=self.device)
        policy_losses = []
        first_update = 1
        if resume:          # not upstream: continue an interrupted run (weights, Adam, RNG, update)
            first_update = self.load_checkpoint(resume) + 1
            global_step = (first_update - 1) * self.batch_size * self.world
            for u in range(1, first_update):         # the expert weight is a running product over the updates already done
                if self.anneal_exp:
                    self.expert_weight *= 1 - ((u - 1) / self.num_updates)
        ck_path, ck_every = self.params_dict.get("checkpoint_path"), int(self.params_dict.get("checkpoint_every") or 0)
        for update in range(first_update, self.num_updates + 1):
            if self.anneal_lr:
                frac = 1.0 - (update - 1.0) / self.num_updates
                self.set_lr(frac * self.learning_rate)
            if self.anneal_exp:
                self.expert_weight *= 1 - ((update - 1) / self.num_updates)
            for step in range(0, self.num_steps):
                global_step += 1 * self.num_envs * self.world
                self.buffer.states[step] = next_state
                self.buffer.observations[step] = next_obs
                self.buffer.terminals[step] = next_done
                next_state, next_obs, next_done = self.rewards_to_go(step, next_state, next_obs, global_step, writer)
            returns, advantages = self.advantages(next_state, next_obs, next_done, self.buffer, self.num_steps)
            buffer = self.buffer.flatten(returns, advantages)
            (policy_loss, value_loss, entropy_loss, old_approx_kl, approx_kl,
             clip_fracs) = self.update(buffer, self.num_update_epochs, self.batch_size, self.minibatch_size, policy_losses)
            policy_losses.append(policy_loss.item())
            var_y = buffer[5].var(unbiased=False)
            ev = float(1 - (buffer[5] - buffer[6]).var(unbiased=False) / var_y)
            explained_var = np.nan if float(var_y) == 0 else ev
            writer.add_scalar("charts/learning_rate", self.get_lr(), global_step)
            writer.add_scalar("losses/value_loss", value_loss.item(), global_step)
            writer.add_scalar("losses/policy_loss", policy_loss.item(), global_step)
            writer.add_scalar("losses/entropy", entropy_loss.item(), global_step)
            writer.add_scalar("losses/old_approx_kl", old_approx_kl.item(), global_step)
            writer.add_scalar("losses/approx_kl", approx_kl.item(), global_step)
            writer.add_scalar("losses/clipfrac", np.mean(clip_fracs), global_step)
            writer.add_scalar("losses/explained_variance", explained_var, global_step)
            writer.add_scalar("charts/SPS", int(global_step / (time.time() - start_time)), global_step)
            if ck_path and ck_every > 0 and update % ck_every == 0 and self.rank == 0:
                self.save_checkpoint(ck_path, update=update)      # mid-run: what --resume continues from
        self.envs.close()
        writer.close()
        if self.save_file_path is not None and self.rank == 0:
            # upstream's keys (actor_state / critic_state / optimizer_state) plus actor_logstd, trainer_state, update
            self.save_checkpoint(self.save_file_path + "actor_critic_" + str(self.num_layers) + ".pt", update=self.num_updates)
        return self.total_returns, self.total_episode_lengths, self.x_indices

    def moving_average(self, data, window_size):
        return np.convolve(data, np.ones(window_size) / window_size, mode="valid")
