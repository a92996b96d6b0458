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
    capturable = True          # step() is a fixed sequence of device ops: a whole rollout can be captured as one hipGraph

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

    def generators(self):
        """Generators step() draws from (a graph capture has to know them)."""
        return [self.gen]

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


class SyntheticArmEnv:
    """Device-resident stand-in for BulletArm's ``EnvWrapper`` (src/utils/env_wrapper.py:7-59) with its
    interface: ``reset() -> (states, obs)``, ``step(actions, auto_reset=False) -> (states, obs, rewards,
    dones)``, ``getNextAction() -> plan (N, 5)``.  Observations are (N, C, H, W) heightmap-like images (C = 1 upstream),
    ``states`` the gripper open/closed flag.  Used when bulletarm is not installed (it is not part of
    this image) and for the synthetic image-observation configs of BASELINE.json."""
    device_native = True

    def __init__(self, num_envs, device, obs_size=128, seed=4321, p_done=0.02, p_reward=0.05, channels=1):
        self.num_envs, self.device, self.obs_size, self.channels = num_envs, torch.device(device), obs_size, channels
        self.p_done, self.p_reward = p_done, p_reward
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)

    def _draw(self):
        n, dev, g = self.num_envs, self.device, self.gen
        states = (torch.rand(n, device=dev, generator=g) < 0.5).float()
        obs = torch.rand((n, self.channels, self.obs_size, self.obs_size), device=dev, generator=g)
        return states, obs

    def reset(self):
        return self._draw()

    def getNextAction(self):
        n, dev, g = self.num_envs, self.device, self.gen
        plan = torch.randn((n, 5), device=dev, generator=g) * torch.tensor([0.5, 0.02, 0.02, 0.02, 0.4], device=dev)
        plan[:, 0] = (plan[:, 0] > 0).float()
        return plan

    def step(self, actions, auto_reset=False):
        n, dev, g = self.num_envs, self.device, self.gen
        rewards = (torch.rand(n, device=dev, generator=g) < self.p_reward).float()
        dones = (torch.rand(n, device=dev, generator=g) < self.p_done).float()
        states, obs = self._draw()
        return states, obs, rewards, dones

    def close(self):
        pass


def make_arm_envs(gym_id, num_envs, device, params, seed_offset=0):
    """bulletarm's env_factory when importable (same config as src/robot_ppo.py:113-135), else the
    synthetic stand-in."""
    if not str(gym_id).lower().startswith("synthetic"):
        try:
            from bulletarm import env_factory
        except ImportError:
            env_factory = None
        if env_factory is not None:
            env_config = {"workspace": np.array([[0.25, 0.65], [-0.2, 0.2], [0.01, 0.25]]), "max_steps": 100,
                          "obs_size": 128, "fast_mode": True, "action_sequence": "pxyzr",
                          "render": params.get("render", False), "num_objects": 2, "random_orientation": True,
                          "robot": "kuka", "workspace_check": "point", "object_scale_range": (1, 1),
                          "hard_reset_freq": 100, "physics_mode": "fast", "view_type": "camera_center_xyz",
                          "obs_type": "pixel", "view_scale": 1.5, "transparent_bin": True}
            planner_config = {"random_orientation": True, "dpos": 0.02, "drot": 0.19634954084936207}
            return _BulletArmWrapper(env_factory.createEnvs(num_envs, gym_id, env_config, planner_config))
    return SyntheticArmEnv(num_envs, device, obs_size=int(params.get("obs_size", 128)),
                           seed=int(params.get("env_seed", 4321)) + seed_offset, channels=int(params.get("obs_channels", 1)))


class _BulletArmWrapper:
    """``EnvWrapper`` over a real bulletarm runner: numpy in, float tensors out."""
    device_native = False

    def __init__(self, envs):
        self.envs = envs

    def reset(self):
        states, _in_hands, obs = self.envs.reset()
        return torch.tensor(states).float(), torch.tensor(obs).float()

    def getNextAction(self):
        return torch.tensor(self.envs.getNextAction()).float()

    def step(self, actions, auto_reset=False):
        (states, _in_hands, obs), rewards, dones = self.envs.step(actions.cpu().numpy(), auto_reset)
        return (torch.tensor(states).float(), torch.tensor(obs).float(), torch.tensor(rewards).float(),
                torch.tensor(dones).float())

    def close(self):
        self.envs.close()
