"""``trainer`` -- the second caller family of the hot path: the ``policyTrainer`` ABC
(src/trainer/policyTrainer.py:5-53), ``bulletTrainer`` (src/trainer/bulletTrainer.py:9-29) and
``ppoBulletTrainer`` (src/trainer/ppoBulletTrainer.py:14-176) driving ``policies.ppoBullet`` -- same class names,
constructor arguments, method names and ``run(simulator, env_config, planner_config, gym_id, actor, critic,
encoder_type)`` entry point.

MI355X-first differences, results unchanged where upstream runs at all:
  * the rollout is kept in dense device tensors (``DenseTransitionBuffer``) written one (num_processes,) row per env
    step, instead of a Python list of per-env namedtuples rebuilt into tensors with ``np.stack`` every update
    (src/utils/buffers.py:64-106; ``ppoBullet._loadBatchToDevice`` still accepts such a list);
  * the update consumes the rollout IN TIME ORDER: upstream draws ``replay_buffer.sample(ppo_batch)`` -- indices with
    replacement, which scrambles the (step, process) layout its own GAE reshape assumes (ppoBulletTrainer.py:168,
    ppoBullet.py:127-130);
  * environments come from ``envs.make_arm_envs`` (bulletarm's runner when installed, else the synthetic stand-in);
    an environment that reports no distance-to-goal simply contributes no shaping term.
"""
from __future__ import annotations

import time
from abc import ABC, abstractmethod

import numpy as np
import torch

from .envs import make_arm_envs
from .policies import ppoBullet
from .robot_ppo import store_returns
from .scalars import make_writer


class policyTrainer(ABC):
    """Template of a policy trainer (src/trainer/policyTrainer.py:5-53)."""

    def __init__(self, track=False, run_id=0):
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.track = track
        self.run_id = run_id

    def set_threads_and_seeds(self, seed=0):
        torch.set_num_threads(torch.get_num_threads())
        torch.manual_seed(seed)
        np.random.seed(seed)
        torch.backends.cudnn.deterministic = True

    @abstractmethod
    def initialize_env(self):
        pass

    @abstractmethod
    def pretrain(self):
        pass

    @abstractmethod
    def evaluate(self):
        pass

    @abstractmethod
    def step_env(self):
        pass

    @abstractmethod
    def run(self):
        pass


class DenseTransitionBuffer:
    """(num_env_steps, num_processes, ...) device tensors with the fields of upstream's ``ExpertTransitionPPO``
    (src/utils/misc.py:7): state, obs, action, reward, done, step_left, expert_action, log_probs, value.  ``add_step``
    stores one row for all processes; ``sample`` returns the filled part flattened time-major (index t*N + n), the
    layout ``ppoBullet``'s GAE reshape expects."""
    FIELDS = ("state", "obs", "action", "reward", "done", "step_left", "expert_action", "log_probs", "value")

    def __init__(self, num_steps, num_processes, obs_shape, action_dim, device):
        z = lambda *s: torch.zeros((num_steps, num_processes) + s, device=device)
        self.state, self.obs = z(), z(*obs_shape)
        self.action, self.expert_action = z(action_dim), z(action_dim)
        self.reward, self.done, self.step_left, self.log_probs, self.value = z(), z(), z(), z(), z()
        self.num_steps, self.num_processes, self.t = num_steps, num_processes, 0

    def __len__(self):
        return self.t * self.num_processes

    def add_step(self, **rows):
        if self.t >= self.num_steps:
            raise IndexError("DenseTransitionBuffer is full: reset() it after each update")
        for k, v in rows.items():
            getattr(self, k)[self.t].copy_(torch.as_tensor(v).reshape(getattr(self, k)[self.t].shape))
        self.t += 1

    def sample(self, batch_size=None):
        t = self.t
        flat = lambda x: x[:t].reshape((t * self.num_processes,) + tuple(x.shape[2:]))
        return {k: flat(getattr(self, k)) for k in self.FIELDS}

    def reset(self):
        self.t = 0


class bulletTrainer(policyTrainer):
    def __init__(self, total_time_steps, num_env_steps, num_processes, save_path=None, aug=False, do_pretraining=True,
                 track=False, run_id=0, transition_type="base"):
        super().__init__(track, run_id)
        self.aug = aug
        self.num_env_steps = num_env_steps
        self.total_time_steps = total_time_steps
        self.replay_buffer = None                    # dense; allocated once the observation shape is known
        self.num_processes = num_processes
        self.do_pretraining = do_pretraining
        self.returns = store_returns(num_processes, 0.99)
        self.eval_returns = store_returns(num_processes, 0.99)
        self.num_eval_processes = 1
        self.save_path = save_path
        self.transition_type = transition_type

    def initialize_env(self, simulator, env_config, planner_config, gym_id):
        cfg = dict(env_config or {})
        params = {"obs_size": cfg.get("obs_size", 128), "obs_channels": cfg.get("obs_channels", 1),
                  "render": cfg.get("render", False), "env_seed": cfg.get("seed", 4321)}
        self.envs = make_arm_envs(gym_id, self.num_processes, self.device, params, 0)
        self.eval_envs = make_arm_envs(gym_id, self.num_eval_processes, self.device, params, 1000)


class ppoBulletTrainer(bulletTrainer):
    def __init__(self, agent: ppoBullet, anneal_lr=False, anneal_exp=False, total_time_steps=100000, num_env_steps=1024,
                 num_processes=5, pretrain_episodes=5000, num_eval_episodes=100, track=False, run_id=0):
        super().__init__(total_time_steps, num_env_steps, num_processes, track=track, run_id=run_id)
        self.agent = agent
        self.anneal_lr = anneal_lr
        self.anneal_exp = anneal_exp
        self.expert_weight = 0.01
        self.ppo_batch = self.num_processes * self.num_env_steps
        self.num_updates = self.total_time_steps // self.ppo_batch
        self.num_eval_episodes = num_eval_episodes
        self.pretrain_episodes = pretrain_episodes
        self.track = track
        self.writer = None
        self.global_step = 0

    # ---- environment plumbing: BulletArm's wrapper returns a distance-to-goal as a fifth value, the stand-in does not
    @staticmethod
    def _step(envs, actions, auto_reset=False):
        out = envs.step(actions, auto_reset=auto_reset)
        return out if len(out) == 5 else (*out, None)

    def pretrain(self, pretrain_episodes):
        """Behaviour cloning on the planner's actions (ppoBulletTrainer.py:28-65): roll the expert until
        ``pretrain_episodes`` episodes have ended, then ten epochs of ``agent.pretrain_update`` over shuffled
        minibatches of 32 (numpy's global stream, as upstream)."""
        if pretrain_episodes == 0:
            return None
        states, obs = self.envs.reset()
        obs_l, exp_l, done_eps = [], [], 0
        while done_eps < pretrain_episodes:
            with torch.no_grad():
                unscaled, scaled = self.agent.getActionFromPlan(self.envs.getNextAction())
                exp_l.append(unscaled.to(self.device))
                obs_l.append(self.agent._tile(obs.to(self.device), states.to(self.device)))
            states, obs, _r, dones, _d = self._step(self.envs, scaled, auto_reset=True)
            done_eps += int(dones.sum().item())
        flat_obs, flat_exp = torch.cat(obs_l), torch.cat(exp_l)
        inds = np.arange(flat_exp.shape[0])
        for _ in range(10):
            np.random.shuffle(inds)
            for index in range(0, len(inds), 32):
                mb = torch.as_tensor(inds[index:index + 32], device=self.device)
                self.agent.pretrain_update(flat_obs[mb], flat_exp[mb])

    def step_env(self, s, o, global_step):
        """One env step for all processes (ppoBulletTrainer.py:67-103): act, ask the planner for the expert action, step,
        record the nine transition fields as one dense row.  A finished episode is reset before the next ``act``: upstream
        calls ``envs.reset_envs(done_idxes)`` and patches the returned state / observation rows (:78-85; it writes the reset
        STATES into the observation rows, a slip -- the reset observations are meant); the wrapper's ``auto_reset=True`` does
        that reset inside ``step`` and returns the patched rows, as ``robot_ppo.rewards_to_go`` uses it."""
        (u_a, a), lp, _m, v = self.agent.act(s.to(self.device), o.to(self.device))
        u_e, _e = self.agent.getActionFromPlan(self.envs.getNextAction())
        n_s, n_o, r, d, dist = self._step(self.envs, a.to(self.device), auto_reset=True)
        if self.replay_buffer is None:
            self.replay_buffer = DenseTransitionBuffer(self.num_env_steps, self.num_processes, tuple(o.shape[1:]),
                                                       u_a.shape[1], self.device)
        self.replay_buffer.add_step(state=s, obs=o, action=u_a, reward=r, done=d, step_left=torch.full_like(r, 100.0),
                                    expert_action=u_e, log_probs=lp, value=v)
        if not getattr(self.envs, "device_native", False):
            for i, rew in enumerate(r.tolist()):
                self.returns.add_value(i, rew)
            for i, dd in enumerate(d.tolist()):
                if dd:
                    ret, length = self.returns.calc_discounted_return(i)
                    self.writer.add_scalar("charts/discounted_episodic_return", ret, global_step)
                    self.writer.add_scalar("charts/episodic_length", length, global_step)
        return n_s, n_o, d, dist

    def evaluate(self, global_step):
        """Mean discounted return of ``num_eval_episodes`` deterministic episodes (ppoBulletTrainer.py:105-137)."""
        s, o = self.eval_envs.reset()
        done_eps, total, rets = 0, 0.0, store_returns(self.num_eval_processes, self.agent.gamma)
        steps = 0
        while done_eps < self.num_eval_episodes and steps < 100 * self.num_eval_episodes:
            (_u, a), _lp, _m, _v = self.agent.act(s.to(self.device), o.to(self.device), deterministic=True)
            s, o, r, d, _dist = self._step(self.eval_envs, a.to(self.device), auto_reset=True)
            steps += 1
            for i, rew in enumerate(r.tolist()):
                rets.add_value(i, rew)
            for i, dd in enumerate(d.tolist()):
                if dd:
                    ret, _len = rets.calc_discounted_return(i)
                    total += ret
                    done_eps += 1
        mean_r = total / max(1, done_eps)
        self.writer.add_scalar("charts/eval_discounted_episodic_return", mean_r, global_step)
        return mean_r

    def run(self, simulator, env_config, planner_config, gym_id, actor, critic, encoder_type, log=True):
        self.initialize_env(simulator, env_config, planner_config, gym_id)
        self.agent.initNet(actor, critic, encoder_type)
        if self.track:
            import wandb
            wandb.init(project="ppo", sync_tensorboard=True, config=None, name="ppo_" + gym_id)
        self.writer = make_writer(f"runs/{gym_id}", write=log)
        self.set_threads_and_seeds(1)
        if self.do_pretraining:
            self.pretrain(self.pretrain_episodes)
        self.evaluate(0)
        start = time.time()
        self.global_step = 0
        n_s, n_o = self.envs.reset()
        n_d = torch.zeros(self.num_processes, device=self.device)
        lr0 = self.agent.actor_lr
        for update in range(1, self.num_updates + 1):
            if self.anneal_lr:        # upstream writes self.optimizer, which does not exist (:153): the actor's is meant
                frac = 1.0 - (update - 1.0) / self.num_updates
                self.agent.pi_optimizer.param_groups[0]["lr"] = frac * lr0
            if self.anneal_exp:
                self.expert_weight *= 1 - ((update - 1) / self.num_updates)
                self.agent.expert_weight = self.expert_weight
            dists = []
            for _step in range(self.num_env_steps):
                self.global_step += self.num_processes
                n_s, n_o, n_d, dist = self.step_env(n_s, n_o, self.global_step)
                if dist is not None:
                    dists.append(torch.as_tensor(dist, device=self.device).reshape(-1))
                if (_step + 1) % 2000 == 0:           # upstream's in-rollout evaluation (ppoBulletTrainer.py:166-168)
                    self.evaluate(_step + 1)
            batch = self.replay_buffer.sample(self.ppo_batch)
            n_o_feed = self.agent._tile(n_o.to(self.device), n_s.to(self.device))
            self.agent.update(batch, n_o_feed, n_d.to(self.device), torch.cat(dists) if dists else None)
            self.replay_buffer.reset()
            if self.global_step % 1000 == 0:          # ppoBulletTrainer.py:177-178
                self.evaluate(self.global_step)
            self.writer.add_scalar("charts/SPS", int(self.global_step / max(time.time() - start, 1e-9)), self.global_step)
        self.envs.close()
        self.writer.close()
        if self.save_path is not None:
            self.agent.save_agent(self.save_path, gym_id)
        return self.agent.last_scalars
