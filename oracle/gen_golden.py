#!/usr/bin/env python
"""Generate tests/golden/*.npz from the REAL reference (build container only).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Runs only where /root/reference exists; the GPU
box never sees the reference, only the committed fixtures.  The reference is imported as-is,
unmodified, under in-memory alias modules:

  * ``nets``  -> {discrete_net, continuous_net, critic} from src/nets/nets.py   (src/ppo.py:3)
  * ``models``-> {actor_critic} from src/models/actor_critic.py                 (src/ppo.py:7)
  * ``gym``   -> object exposing ``vector.SyncVectorEnv`` = a deterministic synthetic vec-env
  * ``torch.utils.tensorboard`` -> recording SummaryWriter

Fixtures are DATA (inputs + the reference's outputs); no reference source text is stored.

    python oracle/gen_golden.py                     # rewrites tests/golden/
    python oracle/gen_golden.py trace evaluate      # only those fixture files
"""
from __future__ import annotations

import hashlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = os.environ.get("AURPPO_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ----------------------------------------------------------------------------- shims
class _Space:
    def __init__(self, shape, n=None):
        self.shape = shape
        self.n = n


class SynthVecEnv:
    """Deterministic vector env with its OWN RandomState (must not touch np.random's global
    stream, which the reference uses for the minibatch shuffle)."""

    current = {}

    def __init__(self, fns):
        c = SynthVecEnv.current
        self.N = len(fns)
        self.D = c["obs_dim"]
        self.continuous = c["continuous"]
        self.A = c["act_dim"]
        self.p_done = c.get("p_done", 0.05)
        self.index_obs = c.get("index_obs", False)
        self.rs = np.random.RandomState(c.get("env_seed", 77))
        self.single_observation_space = _Space((self.D,))
        self.single_action_space = _Space((self.A,), None) if self.continuous else _Space((), self.A)
        self.k = 0

    def _obs(self):
        if self.index_obs:
            return (self.k * self.N + np.arange(self.N, dtype=np.float32)).reshape(self.N, 1)
        return self.rs.standard_normal((self.N, self.D)).astype(np.float32)

    def reset(self, seed=None):
        self.k = 0
        return self._obs(), {}

    def step(self, action):
        self.k += 1
        if self.index_obs:
            c = SynthVecEnv.current
            rew = c["rewards"][self.k - 1]
            done = c["dones"][self.k] if self.k < len(c["dones"]) else c["final_done"]
        else:
            rew = self.rs.standard_normal(self.N).astype(np.float32)
            done = (self.rs.random_sample(self.N) < self.p_done)
        return self._obs(), rew, done, np.zeros(self.N, bool), {}

    def close(self):
        pass


class RecWriter:
    last = None

    def __init__(self, *a, **k):
        self.scalars = []
        RecWriter.last = self

    def add_text(self, *a, **k):
        pass

    def add_scalar(self, tag, val, step):
        self.scalars.append((tag, float(val), int(step)))

    def close(self):
        pass


def load_reference():
    sys.path[:0] = [REF, os.path.join(REF, "src")]
    from src.nets import nets as ref_nets
    from src.models.actor_critic import actor_critic as ref_ac

    m = types.ModuleType("nets")
    m.discrete_net, m.continuous_net, m.critic = ref_nets.discrete_net, ref_nets.continuous_net, ref_nets.critic
    sys.modules["nets"] = m
    m = types.ModuleType("models")
    m.actor_critic = ref_ac
    sys.modules["models"] = m
    g = types.ModuleType("gym")
    g.vector = types.SimpleNamespace(SyncVectorEnv=SynthVecEnv)
    g.make = lambda *a, **k: None
    g.wrappers = types.SimpleNamespace()
    sys.modules["gym"] = g
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = RecWriter
    sys.modules["torch.utils.tensorboard"] = tb
    import src.ppo as ref_ppo
    return ref_ppo, ref_ac


def load_robot_gae():
    """robot_ppo.run_gae (the skip-last variant).  Its module needs attribute-bag
    stubs for packages absent here (never called)."""
    for name in ("bulletarm", "bulletarm.env_factory", "e2cnn", "e2cnn.nn", "e2cnn.gspaces"):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.__path__ = []
            sys.modules[name] = mod
    sys.modules["bulletarm"].env_factory = sys.modules["bulletarm.env_factory"]
    sys.modules["e2cnn"].nn = sys.modules["e2cnn.nn"]
    sys.modules["e2cnn"].gspaces = sys.modules["e2cnn.gspaces"]
    try:
        models = sys.modules["models"]
        if not hasattr(models, "robot_actor_critic"):
            models.robot_actor_critic = object
        import src.robot_ppo as rp
        return rp.robot_ppo.run_gae
    except Exception as e:  # ordinary Python error -> fall back to documenting it
        print("robot_ppo import failed:", repr(e))
        return None


def base_params(**over):
    p = dict(gym_id="Synthetic-v0", seed=1.0, num_steps=128, gae=True, total_timesteps=1024, anneal_lr=True,
             gae_lambda=0.95, num_update_epochs=4, num_envs=4, num_minibatches=4, entropy_coeff=0.01,
             value_coeff=0.5, clip_coeff=0.2, clip_vloss=True, max_grad_norm=0.5, target_kl=None,
             norm_adv=True, capture_video=False, hidden_dim=64, continuous=False, learning_rate=2.5e-4,
             exp_name="golden", num_layers=2, dropout=0.0, gamma=0.99, track=False)
    p.update(over)
    return p


def run_train(ref_ppo, agent):
    """Run the reference's train(); swallow the plot ValueError raised after all results exist
    (src/ppo.py:297,311) and keep its torch.save out of the tree."""
    cwd = os.getcwd()
    real_save = torch.save
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        os.makedirs("plots", exist_ok=True)
        torch.save = lambda *a, **k: None
        try:
            import io
            import contextlib
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                try:
                    agent.train()
                except ValueError as e:
                    if "cannot be empty" not in str(e):
                        raise
        finally:
            torch.save = real_save
            os.chdir(cwd)


# ----------------------------------------------------------------------------- 1. GAE
def gae_big_inputs(ci, T, N):
    rs = np.random.RandomState(1000 + ci)
    r = rs.standard_normal((T, N)).astype(np.float32)
    v = rs.standard_normal((T, N)).astype(np.float32)
    d = (rs.random_sample((T, N)) < 0.02).astype(np.float32)
    nv = rs.standard_normal(N).astype(np.float32)
    nd = (rs.random_sample(N) < 0.02).astype(np.float32)
    return r, v, d, nv, nd


def gen_gae(ref_ppo, robot_gae):
    out = {}
    gcpu = torch.Generator().manual_seed(20240)
    cases = [(128, 4, "rand"), (16, 8, "rand"), (128, 64, "rand"), (5, 3, "rand"), (1, 7, "rand"),
             (16, 8, "all_done"), (16, 8, "none_done"), (16, 8, "done_t0"), (16, 8, "done_last"),
             (16, 8, "next_done"), (128, 1024, "big"), (128, 4096, "big")]
    names = []
    for ci, (T, N, kind) in enumerate(cases):
        if kind == "big":
            # inputs come from a per-case numpy RandomState (a version-stable stream) so the test can
            # regenerate them without shipping 10 MB; see tests/util.py::gae_big_inputs
            r, v, d, nv, nd = (torch.from_numpy(a) for a in gae_big_inputs(ci, T, N))
        else:
            r = torch.randn(T, N, generator=gcpu)
            v = torch.randn(T, N, generator=gcpu)
            d = (torch.rand(T, N, generator=gcpu) < 0.1).float()
            nv = torch.randn(N, generator=gcpu)
            nd = (torch.rand(N, generator=gcpu) < 0.1).float()
        if kind == "all_done":
            d[:] = 1; nd[:] = 1
        elif kind == "none_done":
            d[:] = 0; nd[:] = 0
        elif kind == "done_t0":
            d[:] = 0; d[0] = 1
        elif kind == "done_last":
            d[:] = 0; d[T - 1] = 1
        elif kind == "next_done":
            d[:] = 0; nd[:] = 1
        for (gamma, lam) in ((0.99, 0.95),) if kind == "big" else ((0.99, 0.95), (0.9, 1.0)):
            ns = types.SimpleNamespace(buffer=types.SimpleNamespace(rewards=r, values=v, terminals=d),
                                       num_steps=T, gamma=gamma, gae_lambda=lam)
            ret_g, adv_g = ref_ppo.ppo.run_gae(ns, nv, nd)
            ret_n, adv_n = ref_ppo.ppo.normal_advantage(ns, nv, nd)
            name = f"c{ci}_{kind}_T{T}_N{N}_g{gamma}_l{lam}"
            names.append(name)
            big = kind == "big"
            out[name + "/meta"] = np.array([T, N, gamma, lam], dtype=np.float64)
            if big:
                # inputs are regenerated from the seed by the test; store digests only
                out[name + "/seed"] = np.array([1000 + ci, ci], dtype=np.int64)
                for k, a in (("adv_gae", adv_g), ("ret_gae", ret_g), ("adv_norm", adv_n), ("ret_norm", ret_n)):
                    a = a.numpy()
                    out[f"{name}/{k}_sum"] = np.array([a.astype(np.float64).sum(), (a.astype(np.float64) ** 2).sum()])
                    out[f"{name}/{k}_head"] = a.reshape(-1)[:8].copy()
                    out[f"{name}/{k}_tail"] = a.reshape(-1)[-8:].copy()
                    out[f"{name}/{k}_sha"] = np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)
                # the inputs themselves for exact regeneration independent of torch's RNG stream
                out[name + "/in_sha"] = np.frombuffer(hashlib.sha256(
                    r.numpy().tobytes() + v.numpy().tobytes() + d.numpy().tobytes()).digest(), dtype=np.uint8)
            else:
                for k, a in (("rewards", r), ("values", v), ("terminals", d), ("next_value", nv), ("next_done", nd),
                             ("adv_gae", adv_g), ("ret_gae", ret_g), ("adv_norm", adv_n), ("ret_norm", ret_n)):
                    out[f"{name}/{k}"] = a.numpy().copy()
                if robot_gae is not None and T > 1:
                    import io, contextlib
                    rb = types.SimpleNamespace(rewards=r, values=v, terminals=d)
                    with contextlib.redirect_stdout(io.StringIO()):
                        ret_s, adv_s = robot_gae(ns, nv, nd, rb, T)
                    out[f"{name}/adv_skip"] = adv_s.numpy().copy()
                    out[f"{name}/ret_skip"] = ret_s.numpy().copy()
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "gae.npz"), **out)
    print("gae.npz:", len(names), "cases")


# ----------------------------------------------------------------------------- 2. shuffle
def gen_shuffle():
    out = {}
    for B in (8, 16, 512, 4096, 65536, 131072, 524288):
        np.random.seed(1)                       # src/ppo.py:182
        perms = []
        for upd in range(2):
            b = np.arange(B)                    # src/ppo.py:213
            for ep in range(4):
                np.random.shuffle(b)            # src/ppo.py:217
                perms.append(b.copy())
        st = np.random.get_state()
        out[f"B{B}/state_key"] = st[1].astype(np.uint32)
        out[f"B{B}/state_pos"] = np.array([st[2]], dtype=np.int64)
        if B <= 4096:
            out[f"B{B}/perms"] = np.stack(perms).astype(np.int32)
        else:
            out[f"B{B}/head"] = np.stack([p[:16] for p in perms]).astype(np.int32)
            out[f"B{B}/tail"] = np.stack([p[-16:] for p in perms]).astype(np.int32)
            out[f"B{B}/sha"] = np.stack([np.frombuffer(hashlib.sha256(p.astype(np.int32).tobytes()).digest(),
                                                       dtype=np.uint8) for p in perms])
    np.savez_compressed(os.path.join(OUT, "shuffle.npz"), **out)
    print("shuffle.npz written")


# ----------------------------------------------------------------------------- 3. loss fwd/bwd
class FakePolicy(torch.nn.Module):
    """Stand-in for actor_critic whose per-sample outputs are leaf parameters, so that the
    reference's own loss lines (src/ppo.py:225-264) produce d loss/d{newlogp,newv,entropy} as
    ``.grad`` of those parameters.  obs carries the flat sample index t*N+n."""

    def __init__(self, new_logp, new_v, ent, old_logp, old_v, next_value):
        super().__init__()
        self.p_logp = torch.nn.Parameter(new_logp.clone())
        self.p_v = torch.nn.Parameter(new_v.clone())
        self.p_ent = torch.nn.Parameter(ent.clone())
        self.old_logp, self.old_v, self.next_value = old_logp, old_v, next_value

    def value(self, obs):
        return self.next_value

    def evaluate(self, obs, action=None):
        idx = obs[:, 0].long()
        if action is None:
            return torch.zeros(len(idx)), self.old_logp[idx], None, self.old_v[idx].view(-1, 1)
        return None, self.p_logp[idx], self.p_ent[idx], self.p_v[idx].view(-1, 1)


def _exact_ratio_logs(target_f32):
    """fp32 x such that exp(x) == target in fp32 for torch-CPU AND correctly rounded exp."""
    x0 = np.float32(np.log(np.float64(target_f32)))
    cands = [np.nextafter(x0, np.float32(9), dtype=np.float32) for _ in range(1)]
    xs = [x0]
    lo = x0
    for _ in range(6):
        lo = np.nextafter(lo, np.float32(-9), dtype=np.float32); xs.append(lo)
    hi = x0
    for _ in range(6):
        hi = np.nextafter(hi, np.float32(9), dtype=np.float32); xs.append(hi)
    good = [x for x in sorted(xs)
            if np.float32(np.exp(np.float64(x))) == target_f32
            and torch.exp(torch.tensor([x], dtype=torch.float32))[0].item() == float(target_f32)
            and torch.exp(torch.full((16,), float(x), dtype=torch.float32))[3].item() == float(target_f32)]
    assert good, target_f32
    return good[len(good) // 2]


def gen_loss(ref_ppo):
    out = {}
    names = []
    g = torch.Generator().manual_seed(4242)
    configs = []
    for (T, N) in ((16, 8), (128, 4), (33, 7)):
        for norm_adv in (True, False):
            for clip_vloss in (True, False):
                configs.append((T, N, norm_adv, clip_vloss, "rand"))
    configs += [(16, 8, False, True, "boundary"), (16, 8, True, True, "boundary"), (1, 2, True, True, "rand"),
                (16, 8, True, True, "wide")]
    for ci, (T, N, norm_adv, clip_vloss, kind) in enumerate(configs):
        B = T * N
        scale = 1.0 if kind == "wide" else 0.15
        old_logp = torch.randn(B, generator=g) - 1.0
        new_logp = old_logp + scale * torch.randn(B, generator=g)
        old_v = torch.randn(B, generator=g)
        new_v = old_v + 0.3 * torch.randn(B, generator=g)
        ent = torch.rand(B, generator=g) + 0.5
        rewards = torch.randn(T, N, generator=g)
        dones = (torch.rand(T, N, generator=g) < 0.1).float()
        dones[0] = 0
        final_done = (torch.rand(N, generator=g) < 0.1).float()
        next_value = torch.randn(N, generator=g)
        clip = 0.2
        if kind == "boundary":
            hi32, lo32 = np.float32(1 + clip), np.float32(1 - clip)
            xh, xl = _exact_ratio_logs(hi32), _exact_ratio_logs(lo32)
            # ratio exactly on both clip bounds, exactly 1, just outside; value delta exactly +-clip
            old_logp[:12] = 0.0
            new_logp[:12] = torch.tensor([xh, xl, 0.0, xh, xl, 0.0,
                                          np.nextafter(xh, np.float32(9)), np.nextafter(xl, np.float32(-9)),
                                          xh, xl, 0.5, -0.5], dtype=torch.float32)
            old_v[:12] = torch.tensor([0.5, -0.5, 0.0, 1.0, 2.0, -1.0, 0.25, 0.75, 0.0, 0.0, 0.0, 0.0])
            new_v[:12] = old_v[:12] + torch.tensor([0.2, -0.2, 0.0, np.float32(0.2), np.float32(-0.2), 0.5, -0.5,
                                                    0.1, 0.2, -0.2, 0.0, 0.3], dtype=torch.float32)
        SynthVecEnv.current = dict(obs_dim=1, continuous=False, act_dim=2, index_obs=True,
                                   rewards=rewards.numpy(), dones=dones.numpy().astype(bool),
                                   final_done=final_done.numpy().astype(bool))
        params = base_params(num_steps=T, num_envs=N, total_timesteps=B, num_update_epochs=1, num_minibatches=1,
                             norm_adv=norm_adv, clip_vloss=clip_vloss, max_grad_norm=1e30, anneal_lr=False,
                             entropy_coeff=0.01, value_coeff=0.5, clip_coeff=clip)
        agent = ref_ppo.ppo(params)
        fake = FakePolicy(new_logp, new_v, ent, old_logp, old_v, next_value)
        agent.policy = fake
        agent.optimizer = torch.optim.SGD(fake.parameters(), lr=0.0)   # step leaves params and .grad intact
        cap = {}
        real_adv = agent.advantages

        def adv_hook(next_obs, next_done, _cap=cap, _real=real_adv, _agent=agent):
            ret, adv = _real(next_obs, next_done)
            _cap["returns"], _cap["advantages"] = ret.clone(), adv.clone()
            _cap["terminals"] = _agent.buffer.terminals.clone()
            _cap["rewards"] = _agent.buffer.rewards.clone()
            _cap["values"] = _agent.buffer.values.clone()
            _cap["log_probs"] = _agent.buffer.log_probs.clone()
            _cap["next_done"] = next_done.clone()
            return ret, adv
        agent.advantages = adv_hook
        run_train(ref_ppo, agent)
        sc = {t.split("/")[1]: v for (t, v, s) in RecWriter.last.scalars if t.startswith("losses/")}
        name = f"k{ci}_{kind}_T{T}_N{N}_na{int(norm_adv)}_cv{int(clip_vloss)}"
        names.append(name)
        assert torch.equal(cap["log_probs"].reshape(-1), old_logp) and torch.equal(cap["values"].reshape(-1), old_v)
        out[name + "/meta"] = np.array([T, N, int(norm_adv), int(clip_vloss), clip, 0.01, 0.5], dtype=np.float64)
        for k, a in (("newlogp", new_logp), ("oldlogp", old_logp), ("newv", new_v), ("oldv", old_v), ("entropy", ent),
                     ("adv", cap["advantages"].reshape(-1)), ("ret", cap["returns"].reshape(-1)),
                     ("rewards", cap["rewards"]), ("terminals", cap["terminals"]), ("next_value", next_value),
                     ("next_done", cap["next_done"]),
                     ("g_newlogp", fake.p_logp.grad), ("g_newv", fake.p_v.grad), ("g_entropy", fake.p_ent.grad)):
            out[f"{name}/{k}"] = a.detach().numpy().astype(np.float32).copy()
        out[name + "/scalars"] = np.array([sc["policy_loss"], sc["value_loss"], sc["entropy"], sc["old_approx_kl"],
                                           sc["approx_kl"], sc["clipfrac"]], dtype=np.float64)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **out)
    print("loss.npz:", len(names), "cases")


# ----------------------------------------------------------------------------- 4. end-to-end traces
def gen_trace(ref_ppo):
    out = {}
    names = []
    cfgs = [
        ("cfg1_discrete", dict(obs_dim=4, continuous=False, act_dim=2, env_seed=5),
         dict(num_envs=4, num_steps=128, total_timesteps=3 * 512, continuous=False)),
        ("cfg2_continuous", dict(obs_dim=5, continuous=True, act_dim=3, env_seed=6),
         dict(num_envs=8, num_steps=16, total_timesteps=3 * 128, continuous=True, entropy_coeff=0.0,
              learning_rate=3e-4, num_minibatches=4)),
        ("cfg3_normal_adv_tail", dict(obs_dim=3, continuous=True, act_dim=2, env_seed=7),
         dict(num_envs=5, num_steps=10, total_timesteps=2 * 50, continuous=True, gae=False, num_minibatches=4,
              clip_vloss=False, num_update_epochs=2)),
        # same ragged-tail / discounted-return path, but with the clipped value loss: with clip_vloss=False
        # upstream regresses the critic to its own old values (src/ppo.py:261), a gradient of pure rounding
        # noise, so critic parity is only well-defined in this variant
        ("cfg4_normal_adv_tail_clipv", dict(obs_dim=3, continuous=True, act_dim=2, env_seed=8),
         dict(num_envs=5, num_steps=10, total_timesteps=3 * 50, continuous=True, gae=False, num_minibatches=4,
              clip_vloss=True, num_update_epochs=3)),
        # the other net shapes of -d / -nl (src/run_ppo.py:33,37): what K7w / K8w are held to
        ("cfg5_wide_128x3", dict(obs_dim=12, continuous=True, act_dim=3, env_seed=9),
         dict(num_envs=8, num_steps=16, total_timesteps=3 * 128, continuous=True, num_minibatches=4, hidden_dim=128,
              num_layers=3)),
        ("cfg6_discrete_96x1", dict(obs_dim=6, continuous=False, act_dim=3, env_seed=10),
         dict(num_envs=8, num_steps=32, total_timesteps=3 * 256, continuous=False, num_minibatches=4, hidden_dim=96,
              num_layers=1)),
    ]
    for name, envc, over in cfgs:
        SynthVecEnv.current = envc
        torch.manual_seed(1)
        agent = ref_ppo.ppo(base_params(**over))
        out[f"{name}/params"] = np.array(repr(sorted(base_params(**over).items())))
        init_sd = {k: v.clone() for k, v in agent.policy.state_dict().items()}
        updates = []
        real_adv = agent.advantages

        def adv_hook(next_obs, next_done, _u=updates, _real=real_adv, _agent=agent):
            ret, adv = _real(next_obs, next_done)
            b = _agent.buffer
            _u.append(dict(states=b.states.clone(), actions=b.actions.clone(), log_probs=b.log_probs.clone(),
                           rewards=b.rewards.clone(), terminals=b.terminals.clone(), values=b.values.clone(),
                           next_obs=next_obs.clone(), next_done=next_done.clone(), returns=ret.clone(),
                           advantages=adv.clone(), lr=_agent.optimizer.param_groups[0]["lr"]))
            return ret, adv
        agent.advantages = adv_hook
        perms = []
        real_shuffle = np.random.shuffle

        def rec_shuffle(x, _p=perms):
            real_shuffle(x)
            _p.append(np.array(x, copy=True))
        np.random.shuffle = rec_shuffle
        try:
            run_train(ref_ppo, agent)
        finally:
            np.random.shuffle = real_shuffle
        names.append(name)
        for k, v in init_sd.items():
            out[f"{name}/init/{k}"] = v.numpy().copy()
        for k, v in agent.policy.state_dict().items():
            out[f"{name}/final/{k}"] = v.numpy().copy()
        out[f"{name}/num_updates"] = np.array([len(updates)])
        for u, d in enumerate(updates):
            for k, v in d.items():
                out[f"{name}/u{u}/{k}"] = (v.numpy().copy() if torch.is_tensor(v) else np.array([v]))
        out[f"{name}/perms"] = np.stack(perms).astype(np.int32)
        tags = ["charts/learning_rate", "losses/value_loss", "losses/policy_loss", "losses/entropy",
                "losses/old_approx_kl", "losses/approx_kl", "losses/clipfrac", "losses/explained_variance"]
        sc = RecWriter.last.scalars
        out[f"{name}/scalar_tags"] = np.array(tags)
        out[f"{name}/scalars"] = np.array([[v for (t, v, s) in sc if t == tag] for tag in tags], dtype=np.float64).T
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "trace.npz"), **out)
    print("trace.npz:", names)


# ----------------------------------------------------------------------------- 5. actor_critic.evaluate
def gen_evaluate(ref_ac):
    out = {}
    names = []
    g = torch.Generator().manual_seed(99)
    for name, (D, A, cont, layers, hid) in {"cont_D64_A6": (64, 6, True, 2, 64), "disc_D4_A2": (4, 2, False, 2, 64),
                                            "cont_D5_A3_L3": (5, 3, True, 3, 32),
                                            "cont_D64_A6_H128_L3": (64, 6, True, 3, 128),
                                            "disc_D8_A4_H128_L2": (8, 4, False, 2, 128),
                                            "cont_D128_A6_H96_L1": (128, 6, True, 1, 96)}.items():
        torch.manual_seed(1)
        net = ref_ac(D, (A,) if cont else A, hid, layers, 0.0, cont)
        if cont:
            with torch.no_grad():
                net.actor_logstd.copy_(0.3 * torch.randn(1, A, generator=g))
        obs = torch.randn(37, D, generator=g)
        act = torch.randn(37, A, generator=g) if cont else torch.randint(0, A, (37,), generator=g)
        _, logp, ent, val = net.evaluate(obs, act)
        loss = (logp * torch.linspace(0.5, 1.5, 37)).sum() + 0.3 * ent.sum() + (val.view(-1) ** 2).sum()
        loss.backward()
        names.append(name)
        out[f"{name}/meta"] = np.array([D, A, int(cont), layers, hid])
        for k, v in net.state_dict().items():
            out[f"{name}/sd/{k}"] = v.detach().numpy().copy()
        for k, p in net.named_parameters():
            out[f"{name}/grad/{k}"] = p.grad.numpy().copy()
        out[f"{name}/obs"], out[f"{name}/act"] = obs.numpy(), act.numpy()
        out[f"{name}/logp"], out[f"{name}/ent"], out[f"{name}/val"] = (logp.detach().numpy(), ent.detach().numpy(),
                                                                       val.detach().numpy())
        out[f"{name}/value_fn"] = net.value(obs).detach().numpy()
        # orthogonal init reproducibility: a second construction under the same seed
        torch.manual_seed(1)
        net2 = ref_ac(D, (A,) if cont else A, hid, layers, 0.0, cont)
        out[f"{name}/init_sha"] = np.frombuffer(hashlib.sha256(
            b"".join(v.numpy().tobytes() for v in net2.state_dict().values())).digest(), dtype=np.uint8)
        for k, v in net2.state_dict().items():
            out[f"{name}/init/{k}"] = v.detach().numpy().copy()
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "evaluate.npz"), **out)
    print("evaluate.npz:", names)


# ----------------------------------------------------------------------------- 6. robot_actor_critic
def gen_robot_eval():
    """Non-equivariant robot policy (src/models/robot_actor_critic.py): seeded construction (xavier init
    draws), evaluate / value / decodeActions / getActionFromPlan outputs.  The weights (15 MB) are not
    stored: the test rebuilds them from the same seed and checks their digest."""
    from src.models.robot_actor_critic import robot_actor_critic as ref_rac
    out = {}
    torch.manual_seed(1)
    net = ref_rac(torch.device("cpu"), False)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        net.actor_logstd.copy_(0.2 * torch.randn(1, 5, generator=g))
    state = (torch.rand(4, generator=g) < 0.5).float()
    obs = torch.rand(4, 1, 128, 128, generator=g)
    act = 0.5 * torch.randn(4, 5, generator=g)
    plan = torch.stack([torch.rand(4, generator=g), 0.05 * torch.randn(4, generator=g), 0.05 * torch.randn(4, generator=g),
                        0.05 * torch.randn(4, generator=g), torch.randn(4, generator=g)], 1)
    with torch.no_grad():
        actions, unscaled, logp, ent, val = net.evaluate(state, obs, act)
        v2 = net.value(state, obs)
        u_plan, a_plan = net.getActionFromPlan(plan)
    out["sd_sha"] = np.frombuffer(hashlib.sha256(b"".join(v.numpy().tobytes() for v in net.state_dict().values())).digest(),
                                  dtype=np.uint8)
    out["sd_keys"] = np.array(list(net.state_dict().keys()))
    out["logstd"] = net.actor_logstd.detach().numpy().copy()
    for k, v in (("state", state), ("obs", obs), ("act", act), ("plan", plan), ("actions", actions), ("unscaled", unscaled),
                 ("logp", logp), ("ent", ent), ("val", val), ("value_fn", v2), ("u_plan", u_plan), ("a_plan", a_plan)):
        out[k] = v.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "robot_eval.npz"), **out)
    print("robot_eval.npz written")


def main():
    assert os.path.isdir(REF), f"reference not present at {REF} (fixtures are generated in the build container only)"
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)   # fixed reduction order for the recorded scalars
    ref_ppo, ref_ac = load_reference()
    only = set(sys.argv[1:])          # e.g. `gen_golden.py trace evaluate`: rewrite just those fixture files
    want = lambda k: not only or k in only
    if want("gae"):
        gen_gae(ref_ppo, load_robot_gae())
    if want("shuffle"):
        gen_shuffle()
    if want("loss"):
        gen_loss(ref_ppo)
    if want("trace"):
        gen_trace(ref_ppo)
    if want("evaluate"):
        gen_evaluate(ref_ac)
    if want("robot_eval"):
        gen_robot_eval()


if __name__ == "__main__":
    main()
