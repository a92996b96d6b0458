"""``policies`` API surface of the reference: the ``policy`` ABC (src/policies/policy.py:4-53),
``bulletArmPolicy`` (src/policies/bulletArmPolicy.py: batch -> tensors with the ``/255*0.4`` pixel
scaling, ``decodeActions``, ``getActionFromPlan``) and ``ppoBullet`` (src/policies/ppoBullet.py: the
two-optimizer PPO variant driven by ``trainer/ppoBulletTrainer``), same names and signatures.

Upstream's ``ppoBullet`` is not runnable as written (its CLI branch is dead, the GAE
loop body is dedented out of the loop, ``normal_advantage`` names undefined variables, the policy
loss re-samples actions instead of evaluating the stored ones), so it is API shape, not a numerics
oracle.  ``update`` here implements what those lines intend -- GAE over the (steps, processes)
batch, per-minibatch clipped-surrogate step on ``pi`` then value step on ``critic`` with separate
Adam optimizers and ``clip_grad_value_(1.0)``, KL early stop -- on the same HIP kernels as ``ppo``
(K1 for the advantages, K4+K5 for both gradients in one pass)."""
from __future__ import annotations

import sys
from abc import ABC, abstractmethod

import numpy as np
import torch
from torch import nn


class policy(ABC):
    def __init__(self):
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.loss_calc_dict = {}

    @abstractmethod
    def load_info(self):
        pass

    @abstractmethod
    def _loadBatchToDevice(self):
        pass

    @abstractmethod
    def initNet(self):
        pass

    @abstractmethod
    def update(self):
        pass

    @abstractmethod
    def act(self):
        pass

    @abstractmethod
    def save_agent(self):
        pass


class bulletArmPolicy(policy):
    def __init__(self, dr=8, dx=0.05, dy=0.05, dz=0.05, n_a=5, obs_type="pixel"):
        super().__init__()
        self.n_a = n_a
        self.p_range = torch.tensor([0, 1])
        self.dtheta_range = torch.tensor([-np.pi / dr, np.pi / dr])
        self.dx_range = torch.tensor([-dx, dx])
        self.dy_range = torch.tensor([-dy, dy])
        self.dz_range = torch.tensor([-dz, dz])
        self.obs_type = obs_type

    @staticmethod
    def _stack(batch, field):
        return torch.as_tensor(np.stack([np.asarray(getattr(d, field)) for d in batch]))

    def _loadBatchToDevice(self, batch, device=None):
        """Transitions (namedtuples with state, obs, action, reward, next_state, next_obs, done,
        step_left, expert) -> tensors; pixel observations are scaled ``/255*0.4`` (:21-54)."""
        device = self.device if device is None else device
        st = self._stack
        states = st(batch, "state").long().to(device)
        obs = st(batch, "obs").to(device)
        obs = obs.unsqueeze(1) if obs.dim() == 3 else obs
        actions = st(batch, "action").to(device)
        rewards = torch.as_tensor(np.stack([np.asarray(d.reward).squeeze() for d in batch])).to(device)
        next_states = st(batch, "next_state").long().to(device)
        next_obs = st(batch, "next_obs").to(device)
        next_obs = next_obs.unsqueeze(1) if next_obs.dim() == 3 else next_obs
        non_final = (st(batch, "done").int() ^ 1).float().to(device)
        step_lefts = st(batch, "step_left").to(device)
        is_experts = st(batch, "expert").bool().to(device)
        obs, next_obs = obs / 255 * 0.4, next_obs / 255 * 0.4
        self.loss_calc_dict.update(batch_size=len(batch), states=states, obs=obs, action_idx=actions, rewards=rewards,
                                   next_states=next_states, next_obs=next_obs, non_final_masks=non_final,
                                   step_lefts=step_lefts, is_experts=is_experts)
        return states, obs, actions, rewards, next_states, next_obs, non_final, step_lefts, is_experts

    def load_info(self):
        d = self.loss_calc_dict
        return (d["batch_size"], d["states"], d["obs"], d["action_idx"], d["rewards"], d["next_states"], d["next_obs"],
                d["non_final_masks"], d["step_lefts"], d["is_experts"])

    @staticmethod
    def _tile(obs, states):
        return torch.cat([obs, states.reshape(states.size(0), 1, 1, 1).to(obs.dtype).repeat(1, 1, obs.shape[2], obs.shape[3])], dim=1)

    def _loadLossCalcDict(self):
        bs, states, obs, a, r, ns, nobs, nf, sl, ie = self.load_info()
        if self.obs_type == "pixel":
            obs, nobs = self._tile(obs, states), self._tile(nobs, ns)
        return bs, states, obs, a, r, ns, nobs, nf, sl, ie

    def decodeActions(self, *args):
        rngs = [self.p_range, self.dx_range, self.dy_range, self.dz_range] + ([self.dtheta_range] if self.n_a == 5 else [])
        cols = list(args[:len(rngs)])
        actions = torch.stack([0.5 * (u + 1) * (r[1] - r[0]).to(u.device) + r[0].to(u.device) for u, r in zip(cols, rngs)], dim=1)
        return torch.stack(cols, dim=1), actions

    def getActionFromPlan(self, plan):
        rngs = [self.p_range, self.dx_range, self.dy_range, self.dz_range] + ([self.dtheta_range] if self.n_a == 5 else [])
        cols = []
        for i, r in enumerate(rngs):
            r = r.to(plan.device)
            cols.append(2 * (plan[:, i].clamp(r[0], r[1]) - r[0]) / (r[1] - r[0]) - 1)
        return self.decodeActions(*cols)


class ppoBullet(bulletArmPolicy):
    def __init__(self, alpha=1e-2, actor_lr=1e-3, critic_lr=1e-3, alpha_lr=1e-3, gamma=0.99, gae=True,
                 target_update_frequency=1, num_processes=5, total_steps=10000, update_epochs=10, clip_coeff=0.2,
                 max_grad_norm=0.5, value_coeff=0.5, expert_weight=0.01, entropy_coeff=0.01, gae_lambda=0.95,
                 clip_vloss=False, norm_adv=True, num_minibatches=32, target_kl=0.01, ops=None):
        super().__init__()
        self.alpha, self.actor_lr, self.critic_lr, self.alpha_lr = alpha, actor_lr, critic_lr, alpha_lr
        self.target_update_frequency = target_update_frequency
        self.tau = 1e-2
        self.gae, self.gamma, self.gae_lambda = gae, gamma, gae_lambda
        self.minibatch_size = int(num_processes * total_steps) // num_minibatches
        self.num_update_epochs = update_epochs
        self.clip_coeff, self.max_grad_norm = clip_coeff, max_grad_norm
        self.value_coeff, self.entropy_coeff, self.expert_weight = value_coeff, entropy_coeff, expert_weight
        self.clip_vloss, self.norm_adv = clip_vloss, norm_adv
        self.num_processes = num_processes
        self.flattened_buffer, self.transition_dict = {}, {}
        self.target_kl = target_kl
        if ops is None:
            from . import hip_ops as ops
        self.ops = ops
        self.last_scalars = None

    def initNet(self, actor, critic, encoder_type):
        self.pi, self.critic, self.encoder_type = actor, critic, encoder_type
        self.pi_optimizer = torch.optim.Adam([{"params": self.pi.parameters(), "lr": self.actor_lr}])
        self.v_optimizer = torch.optim.Adam(self.critic.parameters(), lr=self.critic_lr)

    def _loadBatchToDevice(self, batch, device=None):
        """PPO transitions carry value / expert_action / log_probs instead of next_* (:54-87);
        ``batch_size`` becomes the number of STEPS (len(batch) / num_processes)."""
        device = self.device if device is None else device
        if isinstance(batch, dict):
            # dense device-resident rollout (trainer.DenseTransitionBuffer.sample): the same fields as stacked tensors,
            # time-major, already where they are needed -- no per-transition namedtuples, no numpy staging
            n = batch["reward"].shape[0]
            g = lambda k: batch[k].to(device)
            states, obs, actions = g("state").long(), g("obs"), g("action")
            obs = obs.unsqueeze(1) if obs.dim() == 3 else obs
            rewards, non_final = g("reward").float(), 1.0 - g("done").float()
            step_lefts, values = g("step_left"), g("value").float()
            expert, log_probs = g("expert_action").float(), g("log_probs").float()
            obs = obs / 255 * 0.4
            self.loss_calc_dict.update(batch_size=int(n / self.num_processes), states=states, obs=obs, actions=actions,
                                       rewards=rewards, non_final_masks=non_final, step_lefts=step_lefts, values=values,
                                       expert_actions=expert, log_probs=log_probs)
            return states, obs, actions, rewards, non_final, step_lefts, values, expert, log_probs
        st = self._stack
        states = st(batch, "state").long().to(device)
        obs = st(batch, "obs").to(device)
        obs = obs.unsqueeze(1) if obs.dim() == 3 else obs
        actions = st(batch, "action").to(device)
        rewards = torch.as_tensor(np.stack([np.asarray(d.reward).squeeze() for d in batch])).float().to(device)
        non_final = (st(batch, "done").int() ^ 1).float().to(device)
        step_lefts = st(batch, "step_left").to(device)
        values = st(batch, "value").float().to(device)
        expert = st(batch, "expert_action").float().to(device)
        log_probs = st(batch, "log_probs").float().to(device)
        obs = obs / 255 * 0.4
        self.loss_calc_dict.update(batch_size=int(len(batch) / self.num_processes), states=states, obs=obs,
                                   actions=actions, rewards=rewards, non_final_masks=non_final, step_lefts=step_lefts,
                                   values=values, expert_actions=expert, log_probs=log_probs)
        return states, obs, actions, rewards, non_final, step_lefts, values, expert, log_probs

    def load_info(self):
        d = self.loss_calc_dict
        return (d["batch_size"], d["states"], d["obs"], d["actions"], d["rewards"], d["non_final_masks"], d["step_lefts"],
                d["values"], d["expert_actions"], d["log_probs"])

    def _loadLossCalcDict(self):
        bs, states, obs, a, r, nf, sl, v, e, lp = self.load_info()
        if self.obs_type == "pixel":
            obs = self._tile(obs, states)
        return bs, states, obs, a, r, nf, sl, v, e, lp

    def get_buffer_values(self, inds, device=None):
        d = self.loss_calc_dict
        states, obs = d["states"][inds], d["obs"][inds]
        return (states, self._tile(obs, states), d["log_probs"][inds], d["actions"][inds], d["expert_actions"][inds],
                d["advantages"][inds], d["returns"][inds], d["values"][inds])

    def _adv(self, next_value, next_done, mode):
        bs, _s, _o, _a, rewards, non_final, _sl, values, _e, _lp = self.load_info()
        T, N = bs, self.num_processes
        f = lambda t: t.reshape(T, N).float().contiguous()
        done = 1.0 - non_final            # the stored mask is (done ^ 1)
        ret, adv = self.ops.gae(f(rewards), f(values.reshape(-1)), f(done), next_value.reshape(-1).float().contiguous(),
                                next_done.reshape(-1).float().contiguous(), self.gamma, self.gae_lambda, mode)
        return ret.reshape(-1), adv.reshape(-1)

    def run_gae(self, next_value, next_done):
        return self._adv(next_value, next_done, self.ops.GAE)

    def normal_advantage(self, next_value, next_done):
        return self._adv(next_value, next_done, self.ops.NORMAL_ADV)

    def advantages(self, next_obs, next_value, next_done):
        with torch.no_grad():
            next_value = self.critic(next_obs).flatten()
            return self.run_gae(next_value, next_done) if self.gae else self.normal_advantage(next_value, next_done)

    def _minibatch(self, mb_inds):
        states, obs, old_lp, actions, expert, adv, ret, values = self.get_buffer_values(mb_inds)
        a, newlogprob, _mean, entropy = self.pi.sample(obs, actions)
        newvalue = self.critic(obs)
        M = newlogprob.shape[0]
        ent_rows = entropy.reshape(M, -1).sum(1)
        vmode = self.ops.VLOSS_CLIPPED if self.clip_vloss else self.ops.VLOSS_RETURNS
        sc, g_lp, g_v, g_e = self.ops.loss_fwd_bwd(
            newlogprob.detach().reshape(-1).contiguous(), old_lp.reshape(-1).contiguous(), adv.contiguous(),
            newvalue.detach().reshape(-1).contiguous(), values.reshape(-1).contiguous(), ret.contiguous(),
            ent_rows.detach().contiguous(), self.clip_coeff, self.entropy_coeff, self.value_coeff, self.norm_adv, vmode)
        return a, expert, newlogprob, entropy, newvalue, sc, g_lp, g_v, g_e

    def compute_loss_pi(self, mb_inds):
        """Policy loss (+ expert MSE) of one minibatch as a differentiable scalar, entropy (per element, as the actor's
        ``sample`` returns it: upstream takes ``.mean()`` over all of it, :263), approx_kl."""
        a, expert, newlogprob, entropy, _nv, sc, g_lp, _gv, _ge = self._minibatch(mb_inds)
        surrogate = (newlogprob.reshape(-1) * g_lp).sum() - (newlogprob.detach().reshape(-1) * g_lp).sum() + sc[self.ops.S_PG]
        loss = surrogate + self.expert_weight * nn.functional.mse_loss(a, expert)
        return loss, entropy, sc[self.ops.S_KL]

    def compute_loss_v(self, mb_inds):
        _a, _e, _lp, _ent, newvalue, sc, _glp, g_v, _ge = self._minibatch(mb_inds)
        nv = newvalue.reshape(-1)
        return (nv * g_v).sum() - (nv.detach() * g_v).sum() + sc[self.ops.S_VL] * self.value_coeff

    @staticmethod
    def _nan_guard(module):
        for name, p in module.named_parameters():
            if p.grad is not None and torch.isnan(p.grad).any():
                print(f"Warning: NaN detected in the gradients of {name}")
                sys.exit()

    def update(self, data, next_obs, next_done, dists=None):
        self._loadBatchToDevice(data)
        if dists is not None:
            self.loss_calc_dict["rewards"] += 1 - dists
        batch_size = self.loss_calc_dict["batch_size"] * self.num_processes
        returns, advantages = self.advantages(next_obs, None, next_done)
        self.loss_calc_dict["returns"], self.loss_calc_dict["advantages"] = returns, advantages
        rows = []
        for _uep in range(self.num_update_epochs):
            b_inds = torch.arange(batch_size, device=returns.device)       # upstream does not shuffle here (:259)
            approx_kl = None
            for index in range(0, batch_size, self.minibatch_size):
                mb_inds = b_inds[index:index + self.minibatch_size]
                loss, entropy, approx_kl = self.compute_loss_pi(mb_inds)
                pi_loss = loss - self.entropy_coeff * entropy.mean()
                self.pi_optimizer.zero_grad()
                pi_loss.backward()
                nn.utils.clip_grad_value_(self.pi.parameters(), clip_value=1.0)
                self.pi_optimizer.step()
                self._nan_guard(self.pi)
                v_loss = self.compute_loss_v(mb_inds)
                self.v_optimizer.zero_grad()
                v_loss.backward()
                nn.utils.clip_grad_value_(self.critic.parameters(), clip_value=1.0)
                self.v_optimizer.step()
                self._nan_guard(self.critic)
                rows.append((pi_loss.detach().item(), v_loss.detach().item(), float(approx_kl)))
            if approx_kl is not None and float(approx_kl) > self.target_kl:
                break
        self.last_scalars = np.array(rows)
        self.loss_calc_dict = {}

    def act(self, states, obs, deterministic=False):
        with torch.no_grad():
            obs = self._tile(obs, states)
            mean = None
            if deterministic:
                _, log_prob, a, _ent = self.pi.sample(obs)
            else:
                a, log_prob, mean, _ent = self.pi.sample(obs)
            val = self.critic(obs)
            a = a.cpu()
            return self.decodeActions(*[a[:, i] for i in range(self.n_a)]), log_prob.cpu(), (mean.cpu() if mean is not None else None), val.cpu()

    def pretrain_update(self, obs, expert):
        a, _lp, _mean, _ent = self.pi.sample(obs)
        expert_loss = nn.functional.mse_loss(a, expert)
        self.pi_optimizer.zero_grad()
        expert_loss.backward()
        nn.utils.clip_grad_norm_(self.pi.parameters(), self.max_grad_norm)
        self.pi_optimizer.step()
        self._nan_guard(self.pi)

    def save_agent(self, path, env="env"):
        torch.save(self.pi.state_dict(), f"{path}/{env}_{self.encoder_type}_agent.pt")
        torch.save(self.critic.state_dict(), f"{path}/{env}_{self.encoder_type}_critic.pt")
