"""CPU restatement of the reference hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Every function cites the reference lines it restates (paths relative to the
upstream repo root).  Arithmetic is fp32 with the reference's association
order; reductions that the reference does with torch's fp32 cascade sum are
done here in fp64 and rounded once (both are within ~1e-7 relative of the exact
value, far inside the 1e-5 parity tolerance).

Parity status: pinned by tests/golden/*.npz (generated from the real reference
by oracle/gen_golden.py) and by numpy.random.RandomState for the shuffle.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32

# vloss_mode values (mirrors include/aurppo.h)
VLOSS_RETURNS = 0    # 0.5*mean((v-R)^2)              robot_ppo.py:390
VLOSS_CLIPPED = 1    # clipped value loss              ppo.py:250-259
VLOSS_OLDVALUES = 2  # 0.5*mean((v-V_old)^2) (F8)      ppo.py:261

GAE_MODE_GAE = 0        # ppo.py:125-142
GAE_MODE_NORMAL = 1     # ppo.py:145-157
GAE_MODE_SKIP_LAST = 2  # robot_ppo.py:224-244 (bootstrap branch dead)


# --------------------------------------------------------------------------
# a4 / a5: advantage estimation
# --------------------------------------------------------------------------
def gae(rewards, values, terminals, next_value, next_done, gamma, lam, mode=GAE_MODE_GAE):
    """Restates ``ppo.run_gae`` (src/ppo.py:125-142), ``ppo.normal_advantage``
    (src/ppo.py:145-157) and the skip-last variant ``robot_ppo.run_gae``
    (src/robot_ppo.py:224-244).  Returns ``(returns, advantages)`` like the reference.

    Association order of the reference, all in fp32 (python-float scalars are cast to
    fp32 when multiplied into an fp32 tensor; ``gamma*gae_lambda`` is folded in fp64 first):
        delta = (r + ((g*nv)*nnt)) - V ;  A = delta + (((g*l)*nnt)*A_next)
        R_t   = r + ((g*nnt)*R_next)                       (normal mode)
    """
    r = np.ascontiguousarray(rewards, dtype=F32)
    v = np.ascontiguousarray(values, dtype=F32)
    d = np.ascontiguousarray(terminals, dtype=F32)
    nv_last = np.ascontiguousarray(next_value, dtype=F32).reshape(-1)
    nd_last = np.ascontiguousarray(next_done, dtype=F32).reshape(-1)
    T, N = r.shape
    g = F32(gamma)
    gl = F32(float(gamma) * float(lam))
    one = F32(1.0)
    if mode == GAE_MODE_NORMAL:
        ret = np.zeros_like(r)
        nxt = None
        for t in range(T - 1, -1, -1):
            if t == T - 1:
                nnt = one - nd_last
                nxt = nv_last
            else:
                nnt = one - d[t + 1]
                nxt = ret[t + 1]
            ret[t] = r[t] + (g * nnt) * nxt
        adv = ret - v
        return ret, adv
    adv = np.zeros_like(r)
    last = np.zeros(N, dtype=F32)
    t_hi = T - 2 if mode == GAE_MODE_SKIP_LAST else T - 1
    for t in range(t_hi, -1, -1):
        if t == T - 1:
            nnt = one - nd_last
            nvals = nv_last
        else:
            nnt = one - d[t + 1]
            nvals = v[t + 1]
        delta = (r[t] + (g * nvals) * nnt) - v[t]
        last = delta + ((gl * nnt) * last)
        adv[t] = last
    ret = adv + v
    return ret, adv


# --------------------------------------------------------------------------
# a6: numpy legacy MT19937 + Fisher-Yates (np.random.seed / np.random.shuffle)
# --------------------------------------------------------------------------
class MT19937:
    """numpy legacy ``RandomState`` bit generator restated (src/ppo.py:182 seeds it,
    src/ppo.py:217 / src/robot_ppo.py:338 shuffle with it).  The algorithm lives in
    numpy (third-party; the reference pins numpy via src/environment.yml) --
    ``init_genrand`` seeding, standard twist/tempering, ``random_interval`` masked
    rejection on 32-bit draws, descending Fisher-Yates.
    """

    N, M = 624, 397

    def __init__(self, seed: int):
        self.seed(seed)

    def seed(self, seed: int):
        mt = [0] * self.N
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, self.N):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.key = np.array(mt, dtype=np.uint32)
        self.pos = self.N

    def _twist(self):
        mt = [int(x) for x in self.key]
        N, M = self.N, self.M
        for k in range(N):
            y = (mt[k] & 0x80000000) | (mt[(k + 1) % N] & 0x7FFFFFFF)
            mt[k] = mt[(k + M) % N] ^ (y >> 1) ^ (0x9908B0DF if (y & 1) else 0)
        self.key = np.array(mt, dtype=np.uint32)
        self.pos = 0

    def random_u32(self) -> int:
        if self.pos == self.N:
            self._twist()
        y = int(self.key[self.pos])
        self.pos += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def interval(self, mx: int) -> int:
        if mx == 0:
            return 0
        mask = mx
        for sh in (1, 2, 4, 8, 16):
            mask |= mask >> sh
        while True:
            v = self.random_u32() & mask
            if v <= mx:
                return v

    def shuffle(self, x: np.ndarray):
        """In-place, like ``np.random.shuffle`` on a 1-d array."""
        n = len(x)
        for i in range(n - 1, 0, -1):
            j = self.interval(i)
            x[i], x[j] = x[j], x[i]

    def get_state(self):
        return self.key.copy(), self.pos


def epoch_permutations(seed_or_rng, batch_size: int, num_epochs: int):
    """Index arrays exactly as one update of the reference sees them: ``b_inds =
    np.arange(B)`` once (src/ppo.py:213) then ``np.random.shuffle(b_inds)`` per epoch on the
    carried array (src/ppo.py:215-217).  Uses numpy's own RandomState (fast path for big B)."""
    rng = seed_or_rng if isinstance(seed_or_rng, np.random.RandomState) else np.random.RandomState(seed_or_rng)
    b = np.arange(batch_size)
    out = []
    for _ in range(num_epochs):
        rng.shuffle(b)
        out.append(b.copy())
    return out


# --------------------------------------------------------------------------
# a7: minibatch gather
# --------------------------------------------------------------------------
def gather(idx, *srcs):
    """``b_x[mb_inds]`` for each flattened buffer stream (src/ppo.py:219-220,225,236,251-257)."""
    idx = np.asarray(idx)
    return [np.ascontiguousarray(np.asarray(s)[idx]) for s in srcs]


# --------------------------------------------------------------------------
# a9 / a10: advantage normalisation + clipped-surrogate loss, forward and backward
# --------------------------------------------------------------------------
def ppo_loss(newlogp, oldlogp, adv, newv, oldv, ret, entropy, clip, ent_coef, vf_coef,
             norm_adv=True, vloss_mode=VLOSS_CLIPPED):
    """Restates src/ppo.py:225-264 (and src/robot_ppo.py:345-398 for the value branch).

    Returns ``(scalars, g_newlogp, g_newv, g_entropy)`` with
    ``scalars = [loss, pg, vl, ent, old_kl, kl, clipfrac, adv_mean, adv_std]`` where ``vl`` is
    the un-weighted ``value_loss`` of ppo.py:259/261 and ``loss = pg - ent_coef*ent + vl*vf_coef``.
    Gradients follow torch autograd conventions (measured, torch 2.10): elementwise
    ``max(a,b)`` splits a tie 0.5/0.5; ``clamp`` passes gradient on the closed interval.
    """
    nl = np.asarray(newlogp, dtype=F32)
    ol = np.asarray(oldlogp, dtype=F32)
    a = np.asarray(adv, dtype=F32)
    v = np.asarray(newv, dtype=F32).reshape(-1)
    vo = np.asarray(oldv, dtype=F32)
    R = np.asarray(ret, dtype=F32)
    H = np.asarray(entropy, dtype=F32)
    M = nl.shape[0]
    invM = 1.0 / M

    lr = nl - ol                                   # ppo.py:226
    ratio = np.exp(lr.astype(np.float64)).astype(F32)   # ppo.py:228 (correctly rounded exp)
    old_kl = F32(np.mean(-lr, dtype=np.float64))                        # ppo.py:232
    kl = F32(np.mean((ratio - F32(1)) - lr, dtype=np.float64))          # ppo.py:233
    clipfrac = F32(np.mean((np.abs(ratio - F32(1.0)) > F32(clip)), dtype=np.float64))  # ppo.py:234

    if norm_adv:                                    # ppo.py:238-239: (x-mean)/(std(ddof=1)+1e-8)
        a64 = a.astype(np.float64)
        mean = F32(a64.mean())
        std = F32(a64.std(ddof=1)) if M > 1 else F32(np.nan)
        an = (a - mean) / (std + F32(1e-8))
    else:
        mean = F32(a.astype(np.float64).mean())
        std = F32(a.astype(np.float64).std(ddof=1)) if M > 1 else F32(np.nan)
        an = a
    lo, hi = F32(1 - clip), F32(1 + clip)           # fp32 roundings of 1-eps / 1+eps
    rc = np.clip(ratio, lo, hi)
    l1 = -an * ratio                                # ppo.py:243
    l2 = -an * rc                                   # ppo.py:244
    pg = F32(np.mean(np.maximum(l1, l2), dtype=np.float64))   # ppo.py:245
    # d max(l1,l2)/d ratio
    in_rng = ((ratio >= lo) & (ratio <= hi)).astype(F32)
    w1 = np.where(l1 > l2, F32(1), np.where(l1 == l2, F32(0.5), F32(0)))
    w2 = F32(1) - w1
    dpg_dratio = (w1 * (-an) + w2 * (-an) * in_rng) * F32(invM)
    g_newlogp = (dpg_dratio * ratio).astype(F32)    # exp backward: grad * output

    c = F32(clip)
    if vloss_mode == VLOSS_CLIPPED:                 # ppo.py:250-259
        du = v - R
        vu = du * du
        dv = v - vo
        dcl = np.clip(dv, -c, c)
        vcl = vo + dcl
        dc = vcl - R
        vc = dc * dc
        vmax = np.maximum(vu, vc)
        vl = F32(0.5) * F32(np.mean(vmax, dtype=np.float64))
        in_v = ((dv >= -c) & (dv <= c)).astype(F32)
        u1 = np.where(vu > vc, F32(1), np.where(vu == vc, F32(0.5), F32(0)))
        u2 = F32(1) - u1
        dvl = (u1 * (F32(2) * du) + u2 * (F32(2) * dc) * in_v) * F32(0.5 * invM)
    elif vloss_mode == VLOSS_RETURNS:               # robot_ppo.py:390
        du = v - R
        vl = F32(0.5) * F32(np.mean(du * du, dtype=np.float64))
        dvl = (F32(2) * du) * F32(0.5 * invM)
    else:                                           # ppo.py:261 (F8: regresses to old values)
        du = v - vo
        vl = F32(0.5) * F32(np.mean(du * du, dtype=np.float64))
        dvl = (F32(2) * du) * F32(0.5 * invM)
    g_newv = (dvl * F32(vf_coef)).astype(F32)

    ent = F32(np.mean(H, dtype=np.float64))         # ppo.py:263
    loss = F32(F32(pg - F32(ent_coef) * ent) + vl * F32(vf_coef))   # ppo.py:264
    g_entropy = np.full(M, F32(-ent_coef * invM), dtype=F32)
    scalars = np.array([loss, pg, vl, ent, old_kl, kl, clipfrac, mean, std], dtype=F32)
    return scalars, g_newlogp, g_newv, g_entropy


# --------------------------------------------------------------------------
# a11: clip_grad_norm_ + Adam (torch semantics, restated for the K6 kernel)
# --------------------------------------------------------------------------
def grad_norm_clip(flat_grads, max_norm):
    """``nn.utils.clip_grad_norm_`` (src/ppo.py:268): total L2 norm, scale by
    ``min(1, max_norm/(norm+1e-6))``.  Returns ``(clipped, norm)``."""
    g = np.asarray(flat_grads, dtype=F32)
    norm = F32(math.sqrt(float(np.sum(g.astype(np.float64) ** 2))))
    coef = F32(max_norm) / (norm + F32(1e-6))
    coef = min(coef, F32(1.0))
    return (g * coef).astype(F32), norm


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-5):
    """torch.optim.Adam single-tensor step (src/ppo.py:80,269; eps=1e-5), fp64 scalars as torch."""
    p = np.asarray(p, dtype=F32); g = np.asarray(g, dtype=F32)
    m = (m + (g - m) * F32(1 - beta1)).astype(F32)                      # exp_avg.lerp_(grad, 1-b1)
    v = (v * F32(beta2) + (g * g) * F32(1 - beta2)).astype(F32)         # mul_(b2).addcmul_(g,g,1-b2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (np.sqrt(v) / F32(math.sqrt(bc2)) + F32(eps)).astype(F32)
    p = (p - F32(lr / bc1) * (m / denom)).astype(F32)
    return p, m, v


# --------------------------------------------------------------------------
# torch-CPU restatements (the NN side needs autograd): a8 model + the whole update
# --------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def make_actor_critic(state_dim, action_dim, hidden_dim=64, num_layers=2, continuous=True):
    """Restates ``actor_critic`` (src/models/actor_critic.py:8-51) over the Tanh MLPs of
    src/nets/nets.py:19-53 (orthogonal init: hidden gain sqrt(2), actor head 0.01, critic head
    1.0, zero bias; ``dropout`` accepted and ignored upstream).  State-dict keys match the
    reference: ``actor.net.{0,2,4}.*``, ``critic.net.*``, ``actor_logstd``."""
    torch = _torch()
    nn = torch.nn

    def layer_init(layer, std=math.sqrt(2), bias_const=0.0):
        nn.init.orthogonal_(layer.weight, std)
        nn.init.constant_(layer.bias, bias_const)
        return layer

    class _mlp(nn.Module):
        def __init__(self, inp, out, head_std):
            super().__init__()
            layers = [layer_init(nn.Linear(int(np.prod(inp)), hidden_dim)), nn.Tanh()]
            for _ in range(num_layers - 1):
                layers += [layer_init(nn.Linear(hidden_dim, hidden_dim)), nn.Tanh()]
            layers.append(layer_init(nn.Linear(hidden_dim, int(np.prod(out))), head_std))
            self.net = nn.Sequential(*layers)

        def forward(self, x):
            return self.net(x)

    class OracleActorCritic(nn.Module):
        def __init__(self):
            super().__init__()
            self.continuous = continuous
            # construction order actor -> critic -> logstd matches actor_critic.py:20-26 (RNG draws)
            self.actor = _mlp(state_dim, action_dim, 0.01)
            self.critic = _mlp(state_dim, 1, 1.0)
            if continuous:
                self.actor_logstd = nn.Parameter(torch.zeros(1, int(np.prod(action_dim))))

        def value(self, state):
            return self.critic(state).flatten()

        def evaluate(self, state, action=None):
            from torch.distributions import Categorical, Normal
            if self.continuous:
                mean = self.actor(state)
                std = torch.exp(self.actor_logstd.expand_as(mean))
                dist = Normal(mean, std)
                if action is None:
                    action = dist.sample()
                return action, dist.log_prob(action).sum(1), dist.entropy().sum(1), self.critic(state)
            dist = Categorical(logits=self.actor(state))
            if action is None:
                action = dist.sample()
            return action, dist.log_prob(action), dist.entropy(), self.critic(state)

    return OracleActorCritic()


def reference_gae_torch(rewards, values, terminals, next_value, next_done, gamma, lam):
    """The reference's literal per-timestep loop of (N,) torch ops (src/ppo.py:125-142); used by
    the cpu_baseline leg so the baseline pays the same launch pattern the reference pays."""
    torch = _torch()
    advantages = torch.zeros_like(rewards)
    lastgaelam = 0
    T = rewards.shape[0]
    for t in reversed(range(T)):
        if t == T - 1:
            nnt = 1.0 - next_done
            nvals = next_value
        else:
            nnt = 1.0 - terminals[t + 1]
            nvals = values[t + 1]
        delta = rewards[t] + gamma * nvals * nnt - values[t]
        advantages[t] = lastgaelam = delta + gamma * lam * nnt * lastgaelam
    return advantages + values, advantages


def reference_update(policy, optimizer, buf, next_obs, next_done, hp, rng, collect=True, stop_after=None, grads_out=None):
    """One update of the reference trainer on CPU torch tensors: bootstrap value + GAE
    (src/ppo.py:159-166), flatten (src/ppo.py:32-39), E epochs of shuffle + minibatch
    clipped-surrogate steps with ``clip_grad_norm_`` and Adam (src/ppo.py:213-273).

    ``buf``: dict of time-major tensors states (T,N,D), actions (T,N[,A]), log_probs, rewards,
    terminals, values (T,N).  ``hp``: dict with gamma, gae_lambda, gae, num_update_epochs,
    num_minibatches, clip_coeff, entropy_coeff, value_coeff, norm_adv, clip_vloss,
    max_grad_norm, target_kl.  ``rng``: numpy RandomState standing in for the global stream.
    Returns dict(returns, advantages, scalars=[per-minibatch 9-vector], perms=[per-epoch]).

    Checker-only extras (defaults = the reference's behaviour): ``grads_out``, a list that receives the flattened
    gradient of every optimizer step as ``loss.backward()`` left it (before ``clip_grad_norm_``); ``stop_after``, return
    after that many optimizer steps (the multi-GPU parity of SURVEY 8e needs step 1 of every shard, nothing more).
    """
    torch = _torch()
    nn = torch.nn
    T, N = buf["rewards"].shape
    B = T * N
    mb = B // hp["num_minibatches"]
    with torch.no_grad():
        next_value = policy.value(next_obs)
        if hp.get("gae", True):
            returns, advantages = reference_gae_torch(buf["rewards"], buf["values"], buf["terminals"],
                                                      next_value, next_done, hp["gamma"], hp["gae_lambda"])
        else:
            r_np, a_np = gae(buf["rewards"].numpy(), buf["values"].numpy(), buf["terminals"].numpy(),
                             next_value.numpy(), next_done.numpy(), hp["gamma"], hp["gae_lambda"],
                             GAE_MODE_NORMAL)
            returns, advantages = torch.from_numpy(r_np), torch.from_numpy(a_np)
    b_obs = buf["states"].reshape((-1,) + tuple(buf["states"].shape[2:]))
    b_logprobs = buf["log_probs"].reshape(-1)
    b_actions = buf["actions"].reshape((-1,) + tuple(buf["actions"].shape[2:]))
    b_adv = advantages.reshape(-1)
    b_ret = returns.reshape(-1)
    b_val = buf["values"].reshape(-1)
    clip = hp["clip_coeff"]
    b_inds = np.arange(B)
    out_scalars, perms = [], []
    approx_kl = None
    n_opt_steps = 0
    for _ep in range(hp["num_update_epochs"]):
        rng.shuffle(b_inds)
        if collect:
            perms.append(b_inds.copy())
        for start in range(0, B, mb):
            mbi = b_inds[start:start + mb]
            _, newlogprob, entropy, newvalue = policy.evaluate(b_obs[mbi], b_actions[mbi])
            log_ratio = newlogprob - b_logprobs[mbi]
            ratio = log_ratio.exp()
            with torch.no_grad():
                old_approx_kl = (-log_ratio).mean()
                approx_kl = ((ratio - 1) - log_ratio).mean()
                clipfrac = ((ratio - 1.0).abs() > clip).float().mean()
            mb_adv = b_adv[mbi]
            adv_mean, adv_std = mb_adv.mean(), (mb_adv.std() if len(mbi) > 1 else torch.tensor(float("nan")))
            if hp.get("norm_adv", True):
                mb_adv = (mb_adv - adv_mean) / (adv_std + 1e-8)
            loss_one = -mb_adv * ratio
            loss_two = -mb_adv * torch.clamp(ratio, 1 - clip, 1 + clip)
            policy_loss = torch.max(loss_one, loss_two).mean()
            newvalue = newvalue.view(-1)
            if hp.get("clip_vloss", True):
                v_un = (newvalue - b_ret[mbi]) ** 2
                v_cl = b_val[mbi] + torch.clamp(newvalue - b_val[mbi], -clip, clip)
                v_cl = (v_cl - b_ret[mbi]) ** 2
                value_loss = 0.5 * torch.max(v_un, v_cl).mean()
            else:
                value_loss = 0.5 * ((newvalue - b_val[mbi]) ** 2).mean()
            entropy_loss = entropy.mean()
            loss = policy_loss - hp["entropy_coeff"] * entropy_loss + value_loss * hp["value_coeff"]
            optimizer.zero_grad()
            loss.backward()
            if grads_out is not None:
                grads_out.append(torch.cat([p.grad.reshape(-1) for p in policy.parameters()]).clone())
            nn.utils.clip_grad_norm_(policy.parameters(), hp["max_grad_norm"])
            optimizer.step()
            if collect:
                out_scalars.append([x.detach().item() for x in (loss, policy_loss, value_loss, entropy_loss,
                                                                old_approx_kl, approx_kl, clipfrac,
                                                                adv_mean, adv_std)])
            n_opt_steps += 1
            if stop_after is not None and n_opt_steps >= stop_after:
                return {"returns": returns, "advantages": advantages,
                        "scalars": np.array(out_scalars, dtype=np.float64), "perms": perms}
        if hp.get("target_kl") is not None and approx_kl > hp["target_kl"]:
            break
    return {"returns": returns, "advantages": advantages,
            "scalars": np.array(out_scalars, dtype=np.float64), "perms": perms}


def reference_robot_update(policy, optimizer, flat, hp, rng, minibatch_size):
    """``robot_ppo.update`` (src/robot_ppo.py:329-408) on CPU torch tensors with the intended (T,N)
    log-prob semantics (upstream's (T,N,A) buffer is ill-defined).  ``flat`` is the tuple of
    ``torch_buffer.flatten``.  Differences from ``ppo``: 5-tuple ``evaluate(state, obs, action)``,
    un-clipped value loss regresses to the returns (:390), ``clip_grad_norm_`` over the actor's
    parameters only (:401), the reported value loss is pre-multiplied by value_coeff (:392).  The
    expert MSE term (:397) is between two buffer tensors and contributes no gradient; omitted."""
    torch = _torch()
    nn = torch.nn
    (b_states, b_obs, b_logprobs, b_actions, b_adv, b_ret, b_val, _b_true) = flat
    B = b_states.shape[0]
    clip = hp["clip_coeff"]
    b_inds = np.arange(B)
    rows = []
    for _ep in range(hp["num_update_epochs"]):
        rng.shuffle(b_inds)
        for start in range(0, B, minibatch_size):
            mbi = b_inds[start:start + minibatch_size]
            _, _, newlogprob, entropy, newvalue = policy.evaluate(b_states[mbi], b_obs[mbi], b_actions[mbi])
            log_ratio = newlogprob - b_logprobs[mbi]
            ratio = log_ratio.exp()
            with torch.no_grad():
                old_kl = (-log_ratio).mean()
                kl = ((ratio - 1) - log_ratio).mean()
                clipfrac = ((ratio - 1.0).abs() > clip).float().mean()
            mb_adv = b_adv[mbi]
            if hp.get("norm_adv", True):
                mb_adv = (mb_adv - mb_adv.mean()) / (mb_adv.std() + 1e-8)
            pg = torch.max(-mb_adv * ratio, -mb_adv * torch.clamp(ratio, 1 - clip, 1 + clip)).mean()
            newvalue = newvalue.view(-1)
            if hp.get("clip_vloss", True):
                v_un = (newvalue - b_ret[mbi]) ** 2
                v_cl = (b_val[mbi] + torch.clamp(newvalue - b_val[mbi], -clip, clip) - b_ret[mbi]) ** 2
                vl = 0.5 * torch.max(v_un, v_cl).mean()
            else:
                vl = 0.5 * ((newvalue - b_ret[mbi]) ** 2).mean()
            ent = entropy.mean()
            loss = pg - hp["entropy_coeff"] * ent + vl * hp["value_coeff"]
            optimizer.zero_grad()
            loss.backward()
            nn.utils.clip_grad_norm_(policy.actor.parameters(), hp["max_grad_norm"])
            optimizer.step()
            rows.append([x.detach().item() for x in (loss, pg, vl, ent, old_kl, kl, clipfrac)])
    return np.array(rows, dtype=np.float64)


def reference_ppobullet_update(pi, critic, pi_opt, v_opt, batch, next_obs, next_done, hp, num_processes):
    """``ppoBullet.update`` (src/policies/ppoBullet.py:240-298 with :123-152, :180-238) on CPU torch tensors, with the
    maths those lines intend where they do not run as written: the GAE recurrence with its body INSIDE the
    time loop (:137-143 are dedented out of it upstream), and the policy loss evaluating the STORED actions
    (``pi.sample(obs, actions)``; :183 re-samples, which makes the ratio meaningless).  Kept as written: no shuffling
    (:259 ``np.arange``), policy step then value step per minibatch with separate Adam optimizers, ``clip_grad_value_``
    at 1.0 (:272,:285), ``entropy.mean()`` over every element (:263), expert MSE added with ``expert_weight`` (:211-213),
    value loss pre-multiplied by ``value_coeff`` (:236), KL early stop on the last minibatch (:293).

    ``batch``: dict of flat time-major tensors (index t*N+n): states (B,), obs (B,C,H,W) already scaled, actions (B,A),
    rewards, dones, values, log_probs (B,), expert (B,A).  Returns rows of (pi_loss, v_loss, approx_kl)."""
    torch = _torch()
    nn = torch.nn
    N = num_processes
    B = batch["rewards"].shape[0]
    T = B // N
    tile = lambda o, s: torch.cat([o, s.reshape(-1, 1, 1, 1).to(o.dtype).repeat(1, 1, o.shape[2], o.shape[3])], dim=1)
    with torch.no_grad():
        next_value = critic(next_obs).flatten()
    ret, adv = gae(batch["rewards"].reshape(T, N).numpy(), batch["values"].reshape(T, N).numpy(),
                   batch["dones"].reshape(T, N).numpy(), next_value.numpy(), next_done.numpy(), hp["gamma"],
                   hp["gae_lambda"], GAE_MODE_GAE if hp.get("gae", True) else GAE_MODE_NORMAL)
    returns, advantages = torch.from_numpy(ret).reshape(-1), torch.from_numpy(adv).reshape(-1)
    clip = hp["clip_coeff"]
    mbs = hp["minibatch_size"]
    rows = []
    for _ep in range(hp["num_update_epochs"]):
        approx_kl = None
        for start in range(0, B, mbs):
            sl = slice(start, start + mbs)
            obs = tile(batch["obs"][sl], batch["states"][sl])
            a, newlogprob, _mean, entropy = pi.sample(obs, batch["actions"][sl])
            log_ratio = newlogprob.reshape(-1) - batch["log_probs"][sl]
            ratio = log_ratio.exp()
            mb_adv = advantages[sl]
            if hp.get("norm_adv", True):
                mb_adv = (mb_adv - mb_adv.mean()) / (mb_adv.std() + 1e-8)
            policy_loss = torch.max(-mb_adv * ratio, -mb_adv * torch.clamp(ratio, 1 - clip, 1 + clip)).mean()
            loss = policy_loss + hp["expert_weight"] * nn.functional.mse_loss(a, batch["expert"][sl])
            with torch.no_grad():
                approx_kl = ((ratio - 1) - log_ratio).mean()
            pi_loss = loss - hp["entropy_coeff"] * entropy.mean()
            pi_opt.zero_grad()
            pi_loss.backward()
            nn.utils.clip_grad_value_(pi.parameters(), clip_value=1.0)
            pi_opt.step()
            newvalue = critic(obs).reshape(-1)
            if hp.get("clip_vloss", False):
                v_un = (newvalue - returns[sl]) ** 2
                v_cl = (batch["values"][sl] + torch.clamp(newvalue - batch["values"][sl], -clip, clip) - returns[sl]) ** 2
                v_loss = 0.5 * torch.max(v_un, v_cl).mean()
            else:
                v_loss = 0.5 * ((newvalue - returns[sl]) ** 2).mean()
            v_loss = v_loss * hp["value_coeff"]
            v_opt.zero_grad()
            v_loss.backward()
            nn.utils.clip_grad_value_(critic.parameters(), clip_value=1.0)
            v_opt.step()
            rows.append((pi_loss.item(), v_loss.item(), approx_kl.item()))
        if approx_kl is not None and approx_kl > hp["target_kl"]:
            break
    return np.array(rows, dtype=np.float64), returns, advantages


# --------------------------------------------------------------------------
# K9's checker: the element-wise tail of a conv block of the robot policy's encoder
# --------------------------------------------------------------------------
def bias_relu_pool2(x, bias=None, scale=None, plane=None):
    """``nn.MaxPool2d(2)(nn.ReLU()(conv_out))`` of src/nets/base_cnns.py:28-45 with the convolution's bias add and, for
    the first block, the tiled gripper-state channel of src/models/robot_actor_critic.py:58-59 written as
    ``scale[b] * plane[c,h,w]`` (convolution is linear in its input channels).  numpy fp32, association
    ``(x + scale*plane) + bias``; an odd trailing row / column is dropped (floor), as torch's pool does."""
    v = np.asarray(x, dtype=F32)
    if plane is not None:
        v = v + np.asarray(scale, dtype=F32).reshape(-1, 1, 1, 1) * np.asarray(plane, dtype=F32).reshape((1,) + v.shape[1:])
    if bias is not None:
        v = v + np.asarray(bias, dtype=F32).reshape(1, -1, 1, 1)
    v = np.maximum(v, F32(0))
    B, C, H, W = v.shape
    v = v[:, :, :H // 2 * 2, :W // 2 * 2].reshape(B, C, H // 2, 2, W // 2, 2)
    return v.max(axis=(3, 5))
