"""Vector environments for the trainer.  The reference steps ``gym.vector.SyncVectorEnv`` on the
host every rollout step (src/ppo.py:66-68,110); gym is not part of this image, and the BASELINE
metric is defined on synthetic rollout tensors, so two built-ins are provided behind the same
``reset(seed=) -> (obs, info)`` / ``step(a) -> (obs, rew, done, trunc, info)`` interface:

* ``SyntheticVecEnv`` -- device-resident N(0,1) observations / rewards, Bernoulli terminals
  (SURVEY section 8d); ``device_native = True`` tells the trainer to skip the host round trip.
* ``CartPoleVecEnv``  -- CartPole-v1 dynamics (Barto-Sutton-Anderson cart-pole as in gym's
  classic_control: Euler, tau 0.02, 12 deg / 2.4 m limits, 500-step truncation) with auto-reset
  and ``final_info`` episode statistics, for the plumbing config.
"""
from __future__ import annotations

import math
import types

import numpy as np
import torch


class _Space:
    def __init__(self, shape, n=None):
        self.shape = tuple(shape)
        self.n = n


class SyntheticVecEnv:
    device_native = True

    def __init__(self, num_envs, obs_dim, act_dim, continuous, device, seed=1234, p_done=0.02):
        self.num_envs = num_envs
        self.device = torch.device(device)
        self.p_done = p_done
        self.obs_shape = tuple(obs_dim) if isinstance(obs_dim, (tuple, list)) else (int(obs_dim),)
        self.single_observation_space = _Space(self.obs_shape)
        self.single_action_space = _Space((int(act_dim),)) if continuous else _Space((), int(act_dim))
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)

    def _obs(self):
        return torch.randn((self.num_envs,) + self.obs_shape, device=self.device, generator=self.gen)

    def reset(self, seed=None):
        return self._obs(), {}

    def step(self, action):
        rew = torch.randn(self.num_envs, device=self.device, generator=self.gen)
        done = (torch.rand(self.num_envs, device=self.device, generator=self.gen) < self.p_done).float()
        return self._obs(), rew, done, None, {}

    def close(self):
        pass


class CartPoleVecEnv:
    device_native = False
    gravity, masscart, masspole, length, force_mag, tau = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    theta_lim, x_lim, max_steps = 12 * 2 * math.pi / 360, 2.4, 500

    def __init__(self, num_envs, seed=0):
        self.num_envs = num_envs
        self.single_observation_space = _Space((4,))
        self.single_action_space = _Space((), 2)
        self.rs = [np.random.RandomState(seed + i) for i in range(num_envs)]
        self.state = np.zeros((num_envs, 4), np.float64)
        self.steps = np.zeros(num_envs, np.int64)
        self.ret = np.zeros(num_envs, np.float64)

    def _reset_one(self, i):
        self.state[i] = self.rs[i].uniform(-0.05, 0.05, size=4)
        self.steps[i] = 0
        self.ret[i] = 0.0

    def reset(self, seed=None):
        if seed is not None:
            seeds = seed if isinstance(seed, (list, tuple)) else [seed + i for i in range(self.num_envs)]
            self.rs = [np.random.RandomState(int(s)) for s in seeds]
        for i in range(self.num_envs):
            self._reset_one(i)
        return self.state.astype(np.float32), {}

    def step(self, action):
        a = np.asarray(action).reshape(-1)
        x, xd, th, thd = self.state.T
        force = np.where(a == 1, self.force_mag, -self.force_mag)
        ct, st = np.cos(th), np.sin(th)
        total_mass = self.masscart + self.masspole
        pml = self.masspole * self.length
        temp = (force + pml * thd * thd * st) / total_mass
        thacc = (self.gravity * st - ct * temp) / (self.length * (4.0 / 3.0 - self.masspole * ct * ct / total_mass))
        xacc = temp - pml * thacc * ct / total_mass
        self.state = np.stack([x + self.tau * xd, xd + self.tau * xacc, th + self.tau * thd, thd + self.tau * thacc], 1)
        self.steps += 1
        self.ret += 1.0
        term = (np.abs(self.state[:, 0]) > self.x_lim) | (np.abs(self.state[:, 2]) > self.theta_lim)
        trunc = self.steps >= self.max_steps
        done = term | trunc
        info = {}
        if done.any():
            finals = [None] * self.num_envs
            for i in np.nonzero(done)[0]:
                finals[i] = {"episode": {"r": float(self.ret[i]), "l": int(self.steps[i])}}
                self._reset_one(i)
            info["final_info"] = finals
        return self.state.astype(np.float32), np.ones(self.num_envs, np.float32), term | trunc, trunc, info

    def close(self):
        pass


def make_vec_env(gym_id, num_envs, continuous, device, params):
    """Synthetic-* ids -> SyntheticVecEnv (obs_dim/act_dim from params, defaults 64/6 continuous,
    4/2 discrete); otherwise gym's SyncVectorEnv when gym is importable (same wrappers as
    src/ppo.py:85-99), else the built-in CartPole for 'CartPole-v1'."""
    if str(gym_id).lower().startswith("synthetic"):
        obs_dim = params.get("obs_dim", 64 if continuous else 4)
        act_dim = params.get("act_dim", 6 if continuous else 2)
        return SyntheticVecEnv(num_envs, obs_dim, act_dim, continuous, device,
                               seed=int(params.get("env_seed", 1234)) + int(params.get("rank", 0)))
    try:
        import gym  # noqa: F401
    except ImportError:
        gym = None
    if gym is not None:
        def thunk():
            env = gym.make(gym_id)
            env = gym.wrappers.RecordEpisodeStatistics(env)
            if continuous:
                env = gym.wrappers.ClipAction(env)
                env = gym.wrappers.NormalizeObservation(env)
                env = gym.wrappers.TransformObservation(env, lambda obs: np.clip(obs, -10, 10))
                env = gym.wrappers.NormalizeReward(env)
                env = gym.wrappers.TransformReward(env, lambda reward: np.clip(reward, -10, 10))
            return env
        return gym.vector.SyncVectorEnv([thunk for _ in range(num_envs)])
    if gym_id == "CartPole-v1":
        return CartPoleVecEnv(num_envs)
    raise RuntimeError(f"gym is not installed and no built-in environment is named {gym_id!r} "
                       "(built-ins: 'CartPole-v1', 'Synthetic-v0')")
