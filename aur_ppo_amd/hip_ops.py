"""Torch-facing wrappers over the C ABI (include/aurppo.h).  Tensors must be CUDA(HIP), fp32 /
int32, contiguous; every call enqueues on torch's CURRENT stream and returns without syncing."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib

GAE, NORMAL_ADV, GAE_SKIP_LAST = 0, 1, 2
VLOSS_RETURNS, VLOSS_CLIPPED, VLOSS_OLDVALUES = 0, 1, 2
S_LOSS, S_PG, S_VL, S_ENT, S_OLD_KL, S_KL, S_CLIPFRAC, S_ADV_MEAN, S_ADV_STD = range(9)
N_SCALARS = 9


_runtime_checked = False


def _lib_or_raise():
    global _runtime_checked
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise RuntimeError("aur_ppo_amd.hip_ops needs a gfx950 GPU: the HIP kernels have no CPU fallback")
    if not _runtime_checked:
        n = lib.aurppo_device_count()
        if n != torch.cuda.device_count():
            raise RuntimeError(f"libaurppo_hip.so sees {n} device(s) but torch sees {torch.cuda.device_count()}: the "
                               "library is bound to a different HIP runtime than torch (import torch before loading it)")
        _runtime_checked = True
    return lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {_lib.load().aurppo_last_error().decode()}")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t, dtype=torch.float32):
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError(f"expected a contiguous CUDA {dtype} tensor, got {t.dtype} on {t.device}, "
                         f"contiguous={t.is_contiguous()}")
    return C.c_void_p(t.data_ptr())


_ws_cache = {}
_ws_retired = []      # outgrown workspaces stay allocated: a captured hipGraph may still point into them


def _workspace(kind, nbytes, device):
    key = (kind, device.index if device.index is not None else torch.cuda.current_device())
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _ws_retired.append(ws)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


# ------------------------------------------------------------------ K1
def gae(rewards, values, terminals, next_value, next_done, gamma, lam, mode=GAE, out=None, log_probs=None, rec=None):
    """``ppo.run_gae`` / ``normal_advantage`` (src/ppo.py:125-157).  Returns (returns, advantages).
    With ``log_probs`` (T,N) and ``rec`` (T*N,4) the kernel also writes the per-sample record
    {old_logp, A, R, V} the packed gather / loss path consumes."""
    lib = _lib_or_raise()
    T, N = rewards.shape
    if values.shape != (T, N) or terminals.shape != (T, N) or next_value.numel() != N or next_done.numel() != N:
        raise ValueError(f"shape mismatch: rewards {tuple(rewards.shape)}, values {tuple(values.shape)}, terminals "
                         f"{tuple(terminals.shape)}, next_value {tuple(next_value.shape)}, next_done {tuple(next_done.shape)}")
    if out is None:
        adv, ret = torch.empty_like(rewards), torch.empty_like(rewards)
    else:
        ret, adv = out
    if rec is not None:
        if log_probs is None or log_probs.shape != (T, N) or rec.numel() != 4 * T * N:
            raise ValueError("gae: rec needs log_probs (T,N) and rec (T*N,4)")
        _check(lib.aurppo_gae_pack_f32(_ptr(rewards), _ptr(values), _ptr(terminals), _ptr(next_value), _ptr(next_done),
                                       _ptr(log_probs), _ptr(adv), _ptr(ret), _ptr(rec), T, N, float(gamma),
                                       float(lam), int(mode), _stream()), "aurppo_gae_pack_f32")
        return ret, adv
    _check(lib.aurppo_gae_f32(_ptr(rewards), _ptr(values), _ptr(terminals), _ptr(next_value), _ptr(next_done),
                              _ptr(adv), _ptr(ret), T, N, float(gamma), float(lam), int(mode), _stream()),
           "aurppo_gae_f32")
    return ret, adv


# ------------------------------------------------------------------ K2
class MT19937:
    """Device-resident twin of numpy's global legacy stream: ``np.random.seed`` (src/ppo.py:182) and
    ``np.random.shuffle`` (src/ppo.py:217)."""

    def __init__(self, seed: int, max_n: int, device=None):
        lib = _lib_or_raise()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.max_n = int(max_n)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib.aurppo_mt19937_create(C.byref(h), int(seed) & 0xFFFFFFFF, self.max_n, _stream()),
                   "aurppo_mt19937_create")
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.load().aurppo_mt19937_destroy(h)
            except Exception:
                pass

    def seed(self, seed: int):
        _check(_lib.load().aurppo_mt19937_seed(self._h, int(seed) & 0xFFFFFFFF, _stream()), "aurppo_mt19937_seed")

    def get_state(self):
        key = np.empty(624, np.uint32)
        pos = C.c_int32()
        _check(_lib.load().aurppo_mt19937_get_state(self._h, key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos),
                                                    _stream()), "aurppo_mt19937_get_state")
        return key, int(pos.value)

    def set_state(self, key, pos):
        key = np.ascontiguousarray(key, dtype=np.uint32)
        assert key.shape == (624,)
        _check(_lib.load().aurppo_mt19937_set_state(self._h, key.ctypes.data_as(C.POINTER(C.c_uint32)), int(pos),
                                                    _stream()), "aurppo_mt19937_set_state")

    def status_into(self, out):
        """``out[0]`` (1-element fp32 device tensor) = 1 if a shuffle ran out of pre-generated draws since the last
        (re)seed (its permutation is invalid), else 0.  No host sync: read it with the per-update scalars."""
        _check(_lib.load().aurppo_mt19937_status_f32(self._h, _ptr(out), _stream()), "aurppo_mt19937_status_f32")
        return out

    def shuffle_(self, idx):
        """In-place ``np.random.shuffle`` of an int32 device vector."""
        if idx.numel() == 0:
            return idx
        _check(_lib.load().aurppo_shuffle_i32(self._h, _ptr(idx, torch.int32), idx.numel(), _stream()),
               "aurppo_shuffle_i32")
        return idx

    def shuffle_epochs(self, n: int, epochs: int, out=None):
        """The ``epochs`` index arrays of one update (arange once, shuffled in place per epoch,
        src/ppo.py:213-217) as an (epochs, n) int32 tensor."""
        if out is None:
            out = torch.empty((epochs, n), dtype=torch.int32, device=self.device)
        _check(_lib.load().aurppo_shuffle_epochs_i32(self._h, _ptr(out, torch.int32), int(n), int(epochs), _stream()),
               "aurppo_shuffle_epochs_i32")
        return out


def arange_i32(n, device):
    out = torch.empty(n, dtype=torch.int32, device=device)
    _check(_lib_or_raise().aurppo_arange_i32(_ptr(out, torch.int32), n, _stream()), "aurppo_arange_i32")
    return out


# ------------------------------------------------------------------ K3
def gather(idx, srcs, outs=None, probe=None):
    """One fused launch for ``[s[idx] for s in srcs]`` (src/ppo.py:219-220,225,236,251-257).
    ``srcs``: flattened buffer tensors (B, ...) sharing dim 0; ``idx``: int32 (M,).
    ``probe``: optional object whose ``begin()``/``end()`` are called immediately around the C call
    (bench.py records HIP events there, so host-side argument marshalling is not timed)."""
    lib = _lib_or_raise()
    M = idx.numel()
    n = len(srcs)
    if outs is None:
        outs = [torch.empty((M,) + tuple(s.shape[1:]), dtype=torch.float32, device=s.device) for s in srcs]
    row = [int(np.prod(s.shape[1:])) if s.dim() > 1 else 1 for s in srcs]
    for s, o, r in zip(srcs, outs, row):
        if o.numel() != M * r:
            raise ValueError(f"gather: destination has {o.numel()} elements, expected {M}*{r}")
    VP = C.c_void_p * n
    src_a = VP(*[s.data_ptr() for s in srcs])
    dst_a = VP(*[o.data_ptr() for o in outs])
    for t in list(srcs) + list(outs):
        _ptr(t)
    row_a = (C.c_int * n)(*row)
    idx_p, st = _ptr(idx, torch.int32), _stream()
    if probe is not None:
        probe.begin()
    rc = lib.aurppo_gather_f32(idx_p, M, src_a, dst_a, row_a, n, st)
    if probe is not None:
        probe.end()
    _check(rc, "aurppo_gather_f32")
    return outs


# ------------------------------------------------------------------ K4 + K5
def loss_fwd_bwd(newlogp, oldlogp, adv, newv, oldv, ret, entropy, clip, ent_coef, vf_coef, norm_adv=True,
                 vloss_mode=VLOSS_CLIPPED, out_scalars=None):
    """src/ppo.py:225-264 forward and backward.  Returns (scalars[9], g_newlogp, g_newv, g_entropy)."""
    lib = _lib_or_raise()
    M = newlogp.numel()
    for t in (oldlogp, adv, newv, oldv, ret, entropy):
        if t.numel() != M:
            raise ValueError(f"loss_fwd_bwd: expected {M} elements, got {t.numel()}")
    dev = newlogp.device
    if out_scalars is None:
        out_scalars = torch.empty(N_SCALARS, dtype=torch.float32, device=dev)
    g_lp, g_v, g_e = (torch.empty(M, dtype=torch.float32, device=dev) for _ in range(3))
    ws = _workspace("loss", lib.aurppo_loss_workspace_bytes(M), dev)
    _check(lib.aurppo_loss_fwd_bwd_f32(_ptr(newlogp), _ptr(oldlogp), _ptr(adv), _ptr(newv), _ptr(oldv), _ptr(ret),
                                       _ptr(entropy), M, float(clip), float(ent_coef), float(vf_coef),
                                       int(bool(norm_adv)), int(vloss_mode), _ptr(out_scalars), _ptr(g_lp), _ptr(g_v),
                                       _ptr(g_e), C.c_void_p(ws.data_ptr()), _stream()), "aurppo_loss_fwd_bwd_f32")
    return out_scalars, g_lp, g_v, g_e


def loss_fwd_bwd_packed(newlogp, newv, entropy, rec, clip, ent_coef, vf_coef, norm_adv=True, vloss_mode=VLOSS_CLIPPED,
                        out_scalars=None):
    """As ``loss_fwd_bwd`` with the old-side inputs as a gathered (M,4) record {old_logp, adv, ret, old_v}."""
    lib = _lib_or_raise()
    M = newlogp.numel()
    if newv.numel() != M or entropy.numel() != M or rec.numel() != 4 * M:
        raise ValueError("loss_fwd_bwd_packed: size mismatch")
    dev = newlogp.device
    if out_scalars is None:
        out_scalars = torch.empty(N_SCALARS, dtype=torch.float32, device=dev)
    g_lp, g_v, g_e = (torch.empty(M, dtype=torch.float32, device=dev) for _ in range(3))
    ws = _workspace("loss", lib.aurppo_loss_workspace_bytes(M), dev)
    _check(lib.aurppo_loss_fwd_bwd_packed_f32(_ptr(newlogp), _ptr(newv), _ptr(entropy), _ptr(rec), M, float(clip),
                                              float(ent_coef), float(vf_coef), int(bool(norm_adv)), int(vloss_mode),
                                              _ptr(out_scalars), _ptr(g_lp), _ptr(g_v), _ptr(g_e),
                                              C.c_void_p(ws.data_ptr()), _stream()), "aurppo_loss_fwd_bwd_packed_f32")
    return out_scalars, g_lp, g_v, g_e


class PPOLossPackedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, newlogp, newv, entropy, rec, clip, ent_coef, vf_coef, norm_adv, vloss_mode, out_scalars):
        sc, g_lp, g_v, g_e = loss_fwd_bwd_packed(newlogp.detach().contiguous(), newv.detach().reshape(-1).contiguous(),
                                                 entropy.detach().contiguous(), rec, clip, ent_coef, vf_coef, norm_adv,
                                                 vloss_mode, out_scalars)
        ctx.save_for_backward(g_lp, g_v, g_e)
        ctx.v_shape = newv.shape
        return sc[S_LOSS].clone()

    @staticmethod
    def backward(ctx, grad_out):
        g_lp, g_v, g_e = ctx.saved_tensors
        return (g_lp * grad_out, (g_v * grad_out).view(ctx.v_shape), g_e * grad_out) + (None,) * 7


def ppo_loss_packed(newlogp, newv, entropy, rec, clip, ent_coef, vf_coef, norm_adv=True, vloss_mode=VLOSS_CLIPPED,
                    out_scalars=None):
    return PPOLossPackedFn.apply(newlogp, newv, entropy, rec, clip, ent_coef, vf_coef, norm_adv, vloss_mode, out_scalars)


class PPOLossFn(torch.autograd.Function):
    """``loss = ppo_loss(newlogp, newvalue, entropy, ...)`` with the reference's semantics; the
    HIP kernel produces the three input gradients in the forward pass, backward only scales them."""

    @staticmethod
    def forward(ctx, newlogp, newv, entropy, oldlogp, adv, oldv, ret, clip, ent_coef, vf_coef, norm_adv, vloss_mode,
                out_scalars):
        nv = newv.reshape(-1)
        sc, g_lp, g_v, g_e = loss_fwd_bwd(newlogp.detach().contiguous(), oldlogp, adv, nv.detach().contiguous(), oldv,
                                          ret, entropy.detach().contiguous(), clip, ent_coef, vf_coef, norm_adv,
                                          vloss_mode, out_scalars)
        ctx.save_for_backward(g_lp, g_v, g_e)
        ctx.v_shape = newv.shape
        return sc[S_LOSS].clone()

    @staticmethod
    def backward(ctx, grad_out):
        g_lp, g_v, g_e = ctx.saved_tensors
        return (g_lp * grad_out, (g_v * grad_out).view(ctx.v_shape), g_e * grad_out) + (None,) * 10


def ppo_loss(newlogp, newv, entropy, oldlogp, adv, oldv, ret, clip, ent_coef, vf_coef, norm_adv=True,
             vloss_mode=VLOSS_CLIPPED, out_scalars=None):
    return PPOLossFn.apply(newlogp, newv, entropy, oldlogp, adv, oldv, ret, clip, ent_coef, vf_coef, norm_adv,
                           vloss_mode, out_scalars)


# ------------------------------------------------------------------ K7
MLP_HIDDEN, MLP_MAX_D, MLP_MAX_A = 64, 64, 16


MLP_WIDE_MAX_HIDDEN, MLP_WIDE_MAX_D, MLP_WIDE_MAX_LAYERS = 128, 128, 3


def mlp_layout(policy, bucket):
    """Float offsets of ``policy`` (an ``actor_critic``: ``num_layers`` Tanh layers of ``hidden_dim``, src/nets/nets.py:19-53)
    inside ``bucket``'s flat buffers, or None if no fused kernel is built for its shape.

    The default shape (two layers of 64, D <= 64) gets K7/K8's layout {w1,b1,w2,b2,w3,b3} x {actor, critic} + actor_logstd;
    every other shape up to three layers of 128 over D <= 128 gets K7w/K8w's (``wide=True``: {w_l, b_l} for l = 0..L per
    net, then actor_logstd)."""
    NL, Hd = getattr(policy, "num_layers", None), getattr(policy, "hidden_dim", None)
    if not hasattr(policy, "continuous") or not isinstance(NL, int) or not isinstance(Hd, int):
        return None
    if not (1 <= NL <= MLP_WIDE_MAX_LAYERS and 1 <= Hd <= MLP_WIDE_MAX_HIDDEN):
        return None
    cont = bool(policy.continuous)
    off, pos = {}, 0
    for p in bucket.params:
        off[id(p)] = pos
        pos += p.numel()
    try:
        seq = []
        for net in (policy.actor.net, policy.critic.net):
            if len(net) != 2 * NL + 1:
                return None
            for li in range(0, 2 * NL + 1, 2):
                seq += [off[id(net[li].weight)], off[id(net[li].bias)]]
        seq.append(off[id(policy.actor_logstd)] if cont else 0)
        D = policy.actor.net[0].weight.shape[1]
        A = policy.actor.net[2 * NL].weight.shape[0]
        for net, out in ((policy.actor.net, A), (policy.critic.net, 1)):
            dims = [D] + [Hd] * NL + [out]
            for l in range(NL + 1):
                if tuple(net[2 * l].weight.shape) != (dims[l + 1], dims[l]):
                    return None
    except (KeyError, AttributeError, IndexError, TypeError):
        return None
    if A > MLP_MAX_A or (not cont and A < 2):
        return None
    # AURPPO_MLP_FORCE_WIDE=1 (diagnostic): the default shape through K7w too, for a same-box A/B of the two kernels
    fast = NL == 2 and Hd == MLP_HIDDEN and D <= MLP_MAX_D and os.environ.get("AURPPO_MLP_FORCE_WIDE") != "1"
    if not fast and D > MLP_WIDE_MAX_D:
        return None
    return dict(offsets=seq, n_params=pos, D=D, A=A, continuous=cont, hidden=Hd, num_layers=NL, wide=not fast)


def k7_variant():
    """Which build of the fused MLP step the library launches for the default 64-64 policy (include/aurppo.h)."""
    return int(_lib_or_raise().aurppo_k7_variant())


def mlp_step_flops(layout, M):
    """Algorithmic FLOPs of one K7 / K7w launch (un-padded): forward of both nets, weight gradients of every
    layer, input gradients of every layer but the first (which needs none)."""
    D, A, Hd, NL = layout["D"], layout["A"], layout.get("hidden", MLP_HIDDEN), layout.get("num_layers", 2)
    total = 0
    for out in (A, 1):
        dims = [D] + [Hd] * NL + [out]
        mats = [dims[l] * dims[l + 1] for l in range(NL + 1)]
        total += 2 * sum(mats) + sum(mats[1:])      # forward + dW of every layer, dX of layers 2..
    return 2 * total * M


def k7w_kernel(hidden, state_dim):
    """Which K7w kernel runs for a net shape: 1 both nets per workgroup (fp32 MFMA), 2 one net (fp32 MFMA), 3 one net (bf16x3 MFMA)."""
    return int(_lib_or_raise().aurppo_k7w_kernel(int(hidden), int(state_dim)))


def k7w_kernel_name(hidden, state_dim, num_layers):
    k = k7w_kernel(hidden, state_dim)
    return {1: f"k_mlpw_step<{num_layers}, true>", 2: f"k_mlpw_step<{num_layers}, false>", 3: f"k_mlpw3_step<{num_layers}>"}[k]


def reload_knobs():
    """Make the library re-read its AURPPO_* environment knobs now (it reads them once per process otherwise)."""
    _check(_lib_or_raise().aurppo_reload_knobs(), "aurppo_reload_knobs")


def mlp_step_issued_bf16_flops(layout, M):
    """bf16 MFMA FLOPs k_mlp_step3 ISSUES per launch (csrc/mlp3.hip): every fp32 product is six bf16 products (bf16x3.h), the
    heads are padded to 16 outputs, the state to a multiple of 16 columns.  Per 32-row tile and wave (net, column half cb):
    v_mfma_f32_32x32x16_bf16 (32 768 FLOP): F1 6 x ceil(D / 16), F2 24, dH2 6, dW2 + dH1 48, dW1 24 (if cb * 32 < D);
    v_mfma_f32_16x16x32_bf16 (16 384 FLOP): head 12, dW3 12.  None for K7w / the fp32-MFMA build (they issue fp32 MFMAs)."""
    if layout.get("wide"):
        return None
    D = layout["D"]
    nks1 = (D + 15) // 16
    n32 = sum(6 * nks1 + 24 + 6 + 48 + (24 if cb * 32 < D else 0) for _net in range(2) for cb in range(2))
    n16 = 4 * 24
    tiles = (M + 31) // 32
    return tiles * (n32 * 32768 + n16 * 16384)


def _wide_only_step(layout, who):
    if layout.get("wide"):
        raise ValueError(f"{who}: only the default 64-64 MLP has the chained minibatch kernels; this policy "
                         f"({layout['num_layers']} x {layout['hidden']}) steps through mlp_ppo_step + clip_adam_")


def pack_records(rec, actions, out=None):
    """(B, 16) packed records {old_logp, A, R, V, action row, 0 ...} for the fused MLP step (pass them as ``rec`` with
    ``actions=None``): a sample's record and action row then share one 64-B line."""
    lib = _lib_or_raise()
    B = rec.shape[0]
    aw = actions.numel() // B
    if rec.numel() != 4 * B or aw * B != actions.numel() or not 1 <= aw <= 12:
        raise ValueError("pack_records: need (B, 4) records and at most 12 action floats per sample")
    if out is None:
        out = torch.empty((B, 16), dtype=torch.float32, device=rec.device)
    _check(lib.aurppo_pack_records_f32(_ptr(rec), _ptr(actions), B, aw, _ptr(out), _stream()), "aurppo_pack_records_f32")
    return out


def _mlp_buffers_ok(obs, actions, rec, layout):
    D, A = layout["D"], layout["A"]
    aw = A if layout.get("continuous", True) else 1
    if obs.shape[-1] != D:
        return False
    if actions is None:                       # packed records
        return aw <= 12 and rec.numel() == obs.shape[0] * 16
    return actions.numel() == obs.shape[0] * aw and rec.numel() == obs.shape[0] * 4


def _optr(t, dtype=torch.float32):
    return _ptr(t, dtype) if t is not None else None


def mlp_ppo_step(obs, actions, rec, idx, flat_param, layout, flat_grad, clip, ent_coef, vf_coef, norm_adv=True,
                 vloss_mode=VLOSS_CLIPPED, out_scalars=None, events=None):
    """K7: gather + evaluate + loss + backward for one minibatch of the MLP actor-critic; overwrites
    ``flat_grad[:n_params]`` and returns the 9 loss scalars.  ``events``: optional pair of
    ``torch.cuda.Event(enable_timing=True)`` recorded right around the main kernel (bench.py)."""
    lib = _lib_or_raise()
    M, D, A, n = idx.numel(), layout["D"], layout["A"], layout["n_params"]
    cont = layout.get("continuous", True)
    if not _mlp_buffers_ok(obs, actions, rec, layout):
        raise ValueError("mlp_ppo_step: buffer shapes do not match the policy")
    if out_scalars is None:
        out_scalars = torch.empty(N_SCALARS, dtype=torch.float32, device=obs.device)
    if layout.get("wide"):
        ws = _workspace("mlp_wide", lib.aurppo_mlp_wide_workspace_bytes(n, layout["hidden"], D), obs.device)
        lay = (C.c_int * len(layout["offsets"]))(*layout["offsets"])
        if events is not None:
            for ev in events:
                ev.record()
        null = C.c_void_p(0)
        _check(lib.aurppo_mlp_wide_ppo_step_f32(
            _ptr(obs), _optr(actions), _ptr(rec), _ptr(idx, torch.int32), M, D, A, int(cont), layout["hidden"],
            layout["num_layers"], _ptr(flat_param), lay, n, _ptr(flat_grad), float(clip), float(ent_coef), float(vf_coef),
            int(bool(norm_adv)), int(vloss_mode), _ptr(out_scalars), C.c_void_p(ws.data_ptr()), _stream(),
            C.c_void_p(events[0].cuda_event) if events is not None else null,
            C.c_void_p(events[1].cuda_event) if events is not None else null), "aurppo_mlp_wide_ppo_step_f32")
        return out_scalars
    ws = _workspace("mlp", lib.aurppo_mlp_workspace_bytes(n), obs.device)
    lay = (C.c_int * 13)(*layout["offsets"])
    args = (_ptr(obs), _optr(actions), _ptr(rec), _ptr(idx, torch.int32), M, D, A, int(cont), MLP_HIDDEN, _ptr(flat_param), lay, n,
            _ptr(flat_grad), float(clip), float(ent_coef), float(vf_coef), int(bool(norm_adv)), int(vloss_mode),
            _ptr(out_scalars), C.c_void_p(ws.data_ptr()), _stream())
    if events is None:
        _check(lib.aurppo_mlp_ppo_step_f32(*args), "aurppo_mlp_ppo_step_f32")
    else:
        for ev in events:          # materialise the hipEvent handles (torch creates them lazily on record())
            ev.record()
        _check(lib.aurppo_mlp_ppo_step_ev_f32(*args, C.c_void_p(events[0].cuda_event), C.c_void_p(events[1].cuda_event)),
               "aurppo_mlp_ppo_step_ev_f32")
    return out_scalars


def mlp_ppo_minibatch(obs, actions, rec, idx, flat_param, layout, flat_grad, clip, ent_coef, vf_coef, norm_adv, vloss_mode,
                      out_scalars, exp_avg, exp_avg_sq, lr_dev, step_dev, max_norm, betas, eps, out_norm, next_idx=None,
                      chained=False):
    """K7 (K7w) + K6b chained: one whole minibatch (src/ppo.py:219-269) -- fused step, clip_grad_norm_ and Adam over the
    same flat bucket -- in three launches.  ``next_idx``: the slice stepped next (its statistics are prepared by
    this call); ``chained=True`` when the previous call named this ``idx`` as its ``next_idx``."""
    lib = _lib_or_raise()
    M, D, A, n = idx.numel(), layout["D"], layout["A"], layout["n_params"]
    cont = layout.get("continuous", True)
    if not _mlp_buffers_ok(obs, actions, rec, layout):
        raise ValueError("mlp_ppo_minibatch: buffer shapes do not match the policy")
    if min(flat_param.numel(), flat_grad.numel(), exp_avg.numel(), exp_avg_sq.numel()) < n:
        raise ValueError("mlp_ppo_minibatch: the flat bucket is smaller than the policy")
    # alignment padding past n_params is left alone: its gradient is never written, so clip and Adam are no-ops there
    if layout.get("wide"):
        # K7w: prepare + step + slab reduce + clip/Adam; with next_idx the optimizer launch leaves the operand copies and the
        # next slice's statistics, and a chained call then skips its prepare launch
        ws = _workspace("mlp_wide", lib.aurppo_mlp_wide_workspace_bytes(n, layout["hidden"], D), obs.device)
        lay = (C.c_int * len(layout["offsets"]))(*layout["offsets"])
        _check(lib.aurppo_mlp_wide_ppo_minibatch_f32(
            _ptr(obs), _optr(actions), _ptr(rec), _ptr(idx, torch.int32), M, D, A, int(cont), layout["hidden"], layout["num_layers"],
            _ptr(flat_param), lay, n, _ptr(flat_grad), float(clip), float(ent_coef), float(vf_coef), int(bool(norm_adv)),
            int(vloss_mode), _ptr(out_scalars), _ptr(exp_avg), _ptr(exp_avg_sq), float(max_norm), _ptr(lr_dev), _ptr(step_dev),
            float(betas[0]), float(betas[1]), float(eps), _ptr(out_norm),
            _ptr(next_idx, torch.int32) if next_idx is not None else None, int(next_idx.numel()) if next_idx is not None else 0,
            int(bool(chained)), C.c_void_p(ws.data_ptr()), _stream()),
            "aurppo_mlp_wide_ppo_minibatch_f32")
        return out_scalars
    ws = _workspace("mlp", lib.aurppo_mlp_workspace_bytes(n), obs.device)
    lay = (C.c_int * 13)(*layout["offsets"])
    _check(lib.aurppo_mlp_ppo_minibatch_f32(
        _ptr(obs), _optr(actions), _ptr(rec), _ptr(idx, torch.int32), M, D, A, int(cont), MLP_HIDDEN, _ptr(flat_param), lay, n,
        _ptr(flat_grad), float(clip), float(ent_coef), float(vf_coef), int(bool(norm_adv)), int(vloss_mode), _ptr(out_scalars),
        _ptr(exp_avg), _ptr(exp_avg_sq), float(max_norm), _ptr(lr_dev), _ptr(step_dev), float(betas[0]), float(betas[1]),
        float(eps), _ptr(out_norm), _ptr(next_idx, torch.int32) if next_idx is not None else None,
        int(next_idx.numel()) if next_idx is not None else 0, int(bool(chained)), C.c_void_p(ws.data_ptr()), _stream()),
        "aurppo_mlp_ppo_minibatch_f32")
    return out_scalars


def mlp_ppo_grad(obs, actions, rec, idx, flat_param, layout, flat_grad, clip, ent_coef, vf_coef, norm_adv, vloss_mode,
                 out_scalars, step_dev, chained=False):
    """First half of a minibatch for one process per GPU: K7 + slab reduce into ``flat_grad[:n_params]`` (as
    ``mlp_ppo_step``), the Adam step count advanced.  The caller all-reduces ``flat_grad`` and calls ``mlp_ppo_apply``."""
    lib = _lib_or_raise()
    M, D, A, n = idx.numel(), layout["D"], layout["A"], layout["n_params"]
    cont = layout.get("continuous", True)
    if not _mlp_buffers_ok(obs, actions, rec, layout):
        raise ValueError("mlp_ppo_grad: buffer shapes do not match the policy")
    _wide_only_step(layout, "mlp_ppo_grad")
    ws = _workspace("mlp", lib.aurppo_mlp_workspace_bytes(n), obs.device)
    lay = (C.c_int * 13)(*layout["offsets"])
    _check(lib.aurppo_mlp_ppo_grad_f32(
        _ptr(obs), _optr(actions), _ptr(rec), _ptr(idx, torch.int32), M, D, A, int(cont), MLP_HIDDEN, _ptr(flat_param), lay, n,
        _ptr(flat_grad), float(clip), float(ent_coef), float(vf_coef), int(bool(norm_adv)), int(vloss_mode), _ptr(out_scalars),
        _ptr(step_dev), int(bool(chained)), C.c_void_p(ws.data_ptr()), _stream()), "aurppo_mlp_ppo_grad_f32")
    return out_scalars


def mlp_ppo_apply(flat_param, flat_grad, exp_avg, exp_avg_sq, layout, lr_dev, step_dev, max_norm, betas, eps, out_norm,
                  grad_scale=1.0, rec=None, next_idx=None):
    """Second half: ``flat_grad *= grad_scale`` (1/world after a SUM all-reduce), ``clip_grad_norm_`` and ``Adam.step`` in
    one launch, which also prepares ``next_idx`` for the next ``mlp_ppo_grad(..., chained=True)``."""
    lib = _lib_or_raise()
    n = layout["n_params"]
    if min(flat_param.numel(), flat_grad.numel(), exp_avg.numel(), exp_avg_sq.numel()) < n:
        raise ValueError("mlp_ppo_apply: the flat bucket is smaller than the policy")
    _wide_only_step(layout, "mlp_ppo_apply")
    ws = _workspace("mlp", lib.aurppo_mlp_workspace_bytes(n), flat_param.device)
    lay = (C.c_int * 13)(*layout["offsets"])
    _check(lib.aurppo_mlp_ppo_apply_f32(
        _ptr(flat_param), _ptr(flat_grad), _ptr(exp_avg), _ptr(exp_avg_sq), lay, n, layout["D"], float(grad_scale), float(max_norm),
        _ptr(lr_dev), _ptr(step_dev), float(betas[0]), float(betas[1]), float(eps), _ptr(out_norm),
        _ptr(rec) if rec is not None else None, (rec.numel() // rec.shape[0]) if rec is not None else 4,
        _ptr(next_idx, torch.int32) if next_idx is not None else None,
        int(next_idx.numel()) if next_idx is not None else 0, C.c_void_p(ws.data_ptr()), _stream()), "aurppo_mlp_ppo_apply_f32")
    return out_norm


def mlp_ppo_apply_parts(flat_param, flat_grad, exp_avg, exp_avg_sq, layout, lr_dev, step_dev, max_norm, betas, eps, out_norm,
                        sq_part, rec=None, next_idx=None):
    """``mlp_ppo_apply`` for a gradient that already is the mean over ranks and comes with partial sums of squares
    (``P2PExchange.allreduce_mean_``): the clip's norm is their sum, nothing is rescaled."""
    lib = _lib_or_raise()
    n = layout["n_params"]
    if min(flat_param.numel(), flat_grad.numel(), exp_avg.numel(), exp_avg_sq.numel()) < n:
        raise ValueError("mlp_ppo_apply_parts: the flat bucket is smaller than the policy")
    _wide_only_step(layout, "mlp_ppo_apply_parts")
    ws = _workspace("mlp", lib.aurppo_mlp_workspace_bytes(n), flat_param.device)
    lay = (C.c_int * 13)(*layout["offsets"])
    _check(lib.aurppo_mlp_ppo_apply_parts_f32(
        _ptr(flat_param), _ptr(flat_grad), _ptr(exp_avg), _ptr(exp_avg_sq), lay, n, layout["D"], _ptr(sq_part, torch.float64),
        int(sq_part.numel()), float(max_norm), _ptr(lr_dev), _ptr(step_dev), float(betas[0]), float(betas[1]), float(eps),
        _ptr(out_norm), _ptr(rec) if rec is not None else None, (rec.numel() // rec.shape[0]) if rec is not None else 4,
        _ptr(next_idx, torch.int32) if next_idx is not None else None,
        int(next_idx.numel()) if next_idx is not None else 0, C.c_void_p(ws.data_ptr()), _stream()), "aurppo_mlp_ppo_apply_parts_f32")
    return out_norm


class P2PExchange:
    """One-shot all-reduce of a small flat gradient over peer memory (include/aurppo.h, csrc/p2p.hip): every rank publishes its
    vector in a buffer the peers have mapped through HIP IPC and reads theirs directly -- one launch, one hop over xGMI, sums
    formed in rank order (bit-identical on every rank).  ``exchange_handles`` is the one collective it needs, at set-up:
    ``fn(bytes) -> [bytes of rank 0, bytes of rank 1, ...]`` (the trainer passes torch.distributed.all_gather_object)."""

    def __init__(self, rank, world, max_floats, device, exchange_handles):
        lib = _lib_or_raise()
        if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
            raise RuntimeError("P2PExchange: HSA_ENABLE_IPC_MODE_LEGACY=0 must be set before the process touches the GPU "
                               "(this driver only supports dmabuf IPC)")
        self.rank, self.world, self.n_cap, self.device = int(rank), int(world), int(max_floats), device
        self._h = C.c_void_p()
        with torch.cuda.device(device):
            nb = lib.aurppo_p2p_handle_bytes()
            mine, err = b"", None
            try:        # a rank that cannot create its buffer still takes part in the handle exchange (with an empty handle):
                _check(lib.aurppo_p2p_create(C.byref(self._h), self.rank, self.world, self.n_cap, _stream()), "aurppo_p2p_create")
                buf = C.create_string_buffer(nb)
                _check(lib.aurppo_p2p_get_handle(self._h, buf), "aurppo_p2p_get_handle")
                mine = bytes(buf.raw)
            except RuntimeError as e:
                err = str(e)
            every = exchange_handles(mine)          # ... so that every rank sees the failure and raises, none waits for ever
            if len(every) != self.world or any(len(h) != nb for h in every):
                raise RuntimeError("P2PExchange: not every rank could publish an exchange buffer"
                                   + (f" (this rank: {err})" if err else ""))
            if self.world > 1:
                _check(lib.aurppo_p2p_open_peers(self._h, C.create_string_buffer(b"".join(every), nb * self.world)),
                       "aurppo_p2p_open_peers")
        self.sq_part = torch.zeros(lib.aurppo_p2p_parts(self.n_cap), dtype=torch.float64, device=device)

    def allreduce_mean_(self, flat, n, step_dev, timeout_s=10.0):
        """``flat[:n]`` <- mean over ranks; leaves the partial sums of squares of the result in ``self.parts(n)``."""
        lib = _lib_or_raise()
        _check(lib.aurppo_p2p_allreduce_mean_f32(self._h, _ptr(flat), int(n), _ptr(step_dev), _ptr(self.sq_part, torch.float64),
                                                 float(timeout_s), _stream()), "aurppo_p2p_allreduce_mean_f32")
        return flat

    def parts(self, n):
        return self.sq_part[:_lib_or_raise().aurppo_p2p_parts(int(n))]

    def status(self):
        """0 = every exchange met its peers; 1 + r = rank r's flag timed out at least once (sticky).  Synchronises the stream."""
        v = C.c_int(0)
        _check(_lib_or_raise().aurppo_p2p_status(self._h, C.byref(v), _stream()), "aurppo_p2p_status")
        return int(v.value)

    def close(self):
        if self._h:
            _lib_or_raise().aurppo_p2p_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mlp_act(obs, noise, flat_param, layout, actions=None, logp=None, value=None):
    """K8: the rollout step of the MLP actor-critic in one launch.  ``noise``: (N,A) standard-normal draws
    (Gaussian head) or (N,) uniform draws (Categorical head); None -> value only.  Outputs may be rows of
    the rollout buffer.  Returns (actions, logp, value)."""
    lib = _lib_or_raise()
    N, D, A = obs.shape[0], layout["D"], layout["A"]
    cont = layout.get("continuous", True)
    dev = obs.device
    if value is None:
        value = torch.empty(N, dtype=torch.float32, device=dev)
    if noise is not None:
        if actions is None:
            actions = torch.empty((N, A) if cont else (N,), dtype=torch.float32, device=dev)
        if logp is None:
            logp = torch.empty(N, dtype=torch.float32, device=dev)
        if noise.numel() != (N * A if cont else N) or actions.numel() != noise.numel() or logp.numel() != N:
            raise ValueError("mlp_act: noise / output shapes do not match the policy")
    if obs.shape[-1] != D or value.numel() != N:
        raise ValueError("mlp_act: obs / value shapes do not match the policy")
    null = C.c_void_p(0)
    if layout.get("wide"):
        ws = _workspace("mlp_wide", lib.aurppo_mlp_wide_workspace_bytes(0, layout["hidden"], layout["D"]), dev)
        lay = (C.c_int * len(layout["offsets"]))(*layout["offsets"])
        _check(lib.aurppo_mlp_wide_act_f32(_ptr(obs), _ptr(noise) if noise is not None else null, N, D, A, int(cont),
                                           layout["hidden"], layout["num_layers"], _ptr(flat_param), lay, layout["n_params"],
                                           _ptr(actions) if noise is not None else null,
                                           _ptr(logp) if noise is not None else null, _ptr(value),
                                           C.c_void_p(ws.data_ptr()), _stream()), "aurppo_mlp_wide_act_f32")
        return actions, logp, value
    lay = (C.c_int * 13)(*layout["offsets"])
    _check(lib.aurppo_mlp_act_f32(_ptr(obs), _ptr(noise) if noise is not None else null, N, D, A, int(cont), MLP_HIDDEN,
                                  _ptr(flat_param), lay, layout["n_params"], _ptr(actions) if noise is not None else null,
                                  _ptr(logp) if noise is not None else null, _ptr(value), _stream()), "aurppo_mlp_act_f32")
    return actions, logp, value


# ------------------------------------------------------------------ K6
def grad_norm_clip_(flat_grads, max_norm, out_norm=None):
    """In-place ``clip_grad_norm_`` over one flat gradient bucket (src/ppo.py:268).  Returns the
    pre-clip norm as a 1-element device tensor."""
    lib = _lib_or_raise()
    n = flat_grads.numel()
    if out_norm is None:
        out_norm = torch.empty(1, dtype=torch.float32, device=flat_grads.device)
    ws = _workspace("clip", lib.aurppo_clip_workspace_bytes(n), flat_grads.device)
    _check(lib.aurppo_grad_norm_clip_f32(_ptr(flat_grads), n, float(max_norm), _ptr(out_norm),
                                         C.c_void_p(ws.data_ptr()), _stream()), "aurppo_grad_norm_clip_f32")
    return out_norm


def clip_adam_(flat_param, flat_grad, exp_avg, exp_avg_sq, lr_dev, step_dev, max_norm, clip_n=None, betas=(0.9, 0.999),
               eps=1e-5, out_norm=None):
    """K6b: ``clip_grad_norm_`` over the first ``clip_n`` elements + ``Adam.step`` over the whole flat
    bucket, two launches.  ``lr_dev`` / ``step_dev`` are 1-element device tensors (step is incremented)."""
    lib = _lib_or_raise()
    n = flat_param.numel()
    if clip_n is None:
        clip_n = n
    if out_norm is None:
        out_norm = torch.empty(1, dtype=torch.float32, device=flat_param.device)
    ws = _workspace("clip", lib.aurppo_clip_workspace_bytes(n), flat_param.device)
    _check(lib.aurppo_clip_adam_f32(_ptr(flat_param), _ptr(flat_grad), _ptr(exp_avg), _ptr(exp_avg_sq), n, int(clip_n),
                                    float(max_norm), _ptr(lr_dev), _ptr(step_dev), float(betas[0]), float(betas[1]),
                                    float(eps), _ptr(out_norm), C.c_void_p(ws.data_ptr()), _stream()),
           "aurppo_clip_adam_f32")
    return out_norm


# ------------------------------------------------------------------ K9
class _BiasReluPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, scale, plane):
        lib = _lib_or_raise()
        x = x.contiguous()
        B, Cc, Hh, Ww = x.shape
        y = torch.empty((B, Cc, Hh // 2, Ww // 2), dtype=torch.float32, device=x.device)
        mask = torch.empty((B, Cc, Hh // 2, Ww // 2), dtype=torch.uint8, device=x.device)
        if plane is not None:
            plane = plane.detach().reshape(Cc, Hh, Ww).contiguous()
            scale = scale.detach().reshape(B).contiguous()
        _check(lib.aurppo_bias_relu_pool2_fwd_f32(_ptr(x.detach()), _optr(bias.detach().contiguous() if bias is not None else None),
                                                  _optr(scale), _optr(plane), _ptr(y), C.c_void_p(mask.data_ptr()), B, Cc, Hh, Ww,
                                                  _stream()), "aurppo_bias_relu_pool2_fwd_f32")
        ctx.save_for_backward(mask, scale if plane is not None else None)
        ctx.shape = (B, Cc, Hh, Ww)
        ctx.has = (bias is not None, plane is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib_or_raise()
        mask, scale = ctx.saved_tensors
        B, Cc, Hh, Ww = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((B, Cc, Hh, Ww), dtype=torch.float32, device=dy.device)
        part = torch.empty((B, Cc), dtype=torch.float32, device=dy.device) if ctx.has[0] else None
        _check(lib.aurppo_bias_relu_pool2_bwd_f32(_ptr(dy), C.c_void_p(mask.data_ptr()), _ptr(dx), _optr(part), B, Cc, Hh, Ww,
                                                  _stream()), "aurppo_bias_relu_pool2_bwd_f32")
        dbias = part.sum(0) if part is not None else None
        dplane = None
        if ctx.has[1]:
            dplane = torch.empty((1, Cc, Hh, Ww), dtype=torch.float32, device=dy.device)
            _check(lib.aurppo_weighted_batch_sum_f32(_ptr(dx), _ptr(scale), _ptr(dplane), B, Cc * Hh * Ww, _stream()),
                   "aurppo_weighted_batch_sum_f32")
        return dx, dbias, None, dplane


def bias_relu_pool2(x, bias=None, scale=None, plane=None):
    """K9: ``max_pool2d(relu(x + bias[c] + scale[b] * plane[c]), 2)`` in one pass (and one for the backward): the tail of a
    convolution block of src/nets/base_cnns.py:28-45; ``scale`` / ``plane`` carry the tiled gripper-state channel of
    src/models/robot_actor_critic.py:58-59 (no gradient flows to ``scale``, the state is an input)."""
    return _BiasReluPool2.apply(x, bias, scale, plane)


# ------------------------------------------------------------------ K10
class _FirstBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, obs, state, weight, bias):
        lib = _lib_or_raise()
        obs = obs.detach().contiguous()
        state = state.detach().reshape(-1).to(torch.float32).contiguous()
        B, Ci, Hh, Ww = obs.shape
        Co = weight.shape[0]
        w = weight.detach().contiguous()
        y = torch.empty((B, Co, Hh // 2, Ww // 2), dtype=torch.float32, device=obs.device)
        mask = torch.empty((B, Co, Hh // 2, Ww // 2), dtype=torch.uint8, device=obs.device)
        _check(lib.aurppo_first_block_fwd_f32(_ptr(obs), _ptr(w), _optr(bias.detach().contiguous() if bias is not None else None),
                                              _ptr(state), _ptr(y), C.c_void_p(mask.data_ptr()), B, Ci, Co, Hh, Ww, _stream()),
               "aurppo_first_block_fwd_f32")
        ctx.save_for_backward(obs, state, mask)
        ctx.dims = (B, Ci, Co, Hh, Ww, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib_or_raise()
        obs, state, mask = ctx.saved_tensors
        B, Ci, Co, Hh, Ww, has_bias = ctx.dims
        G = B * (Co // 16)
        dw_part = torch.empty((G, 16, (Ci + 1) * 9), dtype=torch.float32, device=dy.device)
        db_part = torch.empty((G, 16), dtype=torch.float32, device=dy.device)
        _check(lib.aurppo_first_block_bwd_f32(_ptr(dy.contiguous()), C.c_void_p(mask.data_ptr()), _ptr(obs), _ptr(state),
                                              _ptr(dw_part), _ptr(db_part), B, Ci, Co, Hh, Ww, _stream()),
               "aurppo_first_block_bwd_f32")
        # (B, Co/16, 16, taps) -> sum over the samples -> (Co, Ci+1, 3, 3)
        dw = dw_part.view(B, Co // 16, 16, (Ci + 1) * 9).sum(0).reshape(Co, Ci + 1, 3, 3)
        db = db_part.view(B, Co // 16, 16).sum(0).reshape(Co) if has_bias else None
        return None, None, dw, db


CONV3X3_MIN_PIXELS = 131072
CONV3X3_MIN_WGS = 512


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, pad):
        lib = _lib_or_raise()
        x, w = x.contiguous(), w.contiguous()
        B, Ci, Hh, Ww = x.shape
        Co = w.shape[0]
        z = torch.empty((B, Co, Hh + 2 * pad - 2, Ww + 2 * pad - 2), dtype=torch.float32, device=x.device)
        ws = _workspace("conv", lib.aurppo_conv3x3_wop_bytes(Ci, Co), x.device)
        _check(lib.aurppo_conv3x3_f32(_ptr(x.detach()), _ptr(w.detach()), _ptr(z), B, Ci, Co, Hh, Ww, int(pad), 0,
                                      C.c_void_p(ws.data_ptr()), _stream()), "aurppo_conv3x3_f32")
        ctx.save_for_backward(x, w)
        ctx.pad = int(pad)
        return z

    @staticmethod
    def backward(ctx, g):
        lib = _lib_or_raise()
        x, w = ctx.saved_tensors
        g = g.contiguous()
        B, Ci, Hh, Ww = x.shape
        Co, pad = w.shape[0], ctx.pad
        dx = dw = None
        if ctx.needs_input_grad[0] and Ci % 32 != 0 and os.environ.get("AURPPO_K11_DGRAD16") != "1":
            # an input gradient with 16 output channels would leave half of every 32-wide matrix block empty: the library's kernel
            dx = torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
        elif ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            ws = _workspace("conv", lib.aurppo_conv3x3_wop_bytes(Co, Ci), x.device)
            _check(lib.aurppo_conv3x3_f32(_ptr(g), _ptr(w.detach()), _ptr(dx), B, Ci, Co, g.shape[2], g.shape[3], pad, 1,
                                          C.c_void_p(ws.data_ptr()), _stream()), "aurppo_conv3x3_f32")
        if ctx.needs_input_grad[1]:
            if conv3x3_wgrad_ok(x, Ci, Co, pad):
                dw = conv3x3_wgrad(g, x, Co, pad)       # K12
            else:
                # small channel counts would leave most of K12's 128 x 128 tile empty: the library's implicit-GEMM kernels
                dw = torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1]
        return dx, dw, None


def conv3x3_wgrad_ok(x, cin, cout, pad):
    """K12's shape rule: the product's tile is 128 (64 for cout <= 64) output channels x 128 (192) filter columns (cin * 9), and
    the layers below 64 output channels / 32 input channels leave most of it empty (the library's small-tile kernels win there).
    ``AURPPO_NO_K12=1`` switches it off, ``AURPPO_K12_ALL=1`` takes every shape the kernel accepts (tests)."""
    env = os.environ
    if env.get("AURPPO_NO_K12") == "1" or not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and pad in (0, 1, 2)):
        return False
    if _lib_or_raise().aurppo_conv3x3_wgrad_ws_bytes(x.shape[0], cin, cout, x.shape[2], x.shape[3], pad) == 0:
        return False
    if env.get("AURPPO_K12_ALL") == "1":
        return True
    pixels = x.shape[0] * (x.shape[2] + 2 * pad - 2) * (x.shape[3] + 2 * pad - 2)
    return cout >= K12_MIN_COUT and cin >= K12_MIN_CIN and pixels >= K12_MIN_PIXELS


K12_MIN_COUT, K12_MIN_CIN, K12_MIN_PIXELS = 64, 32, 16384


def conv3x3_wgrad(g, x, cout, pad):
    """K12: the gradient of ``conv2d(x, w (cout, cin, 3, 3), padding=pad)`` with respect to ``w`` given the output gradient ``g``
    (csrc/conv.hip::k_conv3x3_wgrad: a product over the batch's output pixels, both operands split once per workgroup)."""
    lib = _lib_or_raise()
    g, x = g.contiguous(), x.contiguous()
    B, Ci, Hh, Ww = x.shape
    assert g.shape == (B, cout, Hh + 2 * pad - 2, Ww + 2 * pad - 2), (g.shape, x.shape, cout, pad)
    nb = lib.aurppo_conv3x3_wgrad_ws_bytes(B, Ci, cout, Hh, Ww, int(pad))
    if nb == 0:
        raise RuntimeError("aur_ppo_amd: aurppo_conv3x3_wgrad_f32 does not take this shape")
    ws = _workspace("wgrad", nb, x.device)
    dw = torch.empty((cout, Ci, 3, 3), dtype=torch.float32, device=x.device)
    _check(lib.aurppo_conv3x3_wgrad_f32(_ptr(g), _ptr(x.detach()), _ptr(dw), B, Ci, cout, Hh, Ww, int(pad), C.c_void_p(ws.data_ptr()),
                                        _stream()), "aurppo_conv3x3_wgrad_f32")
    return dw


def conv3x3_ok(x, cin, cout, pad):
    """K11's shape rule for ``conv2d(x, w (cout, cin, 3, 3), padding=pad)``, stride 1: a CUDA fp32 NCHW tensor, cin a multiple
    of 16 (one k-step = 16 channels of a tap), cout a multiple of 32 (whole matrix blocks), and enough output pixels -- a workgroup
    takes 256 of them, and below ~512 workgroups (the encoder's last 3x3 -> 1x1 layers) the launch is a handful of long serial K
    loops that the library's kernels beat."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == cin and cin % 16 == 0 and cout % 32 == 0
            and pad in (0, 1, 2)):
        return False
    pixels = x.shape[0] * (x.shape[2] + 2 * pad - 2) * (x.shape[3] + 2 * pad - 2)
    # a workgroup takes 256 output pixels x 128 output channels: the launch needs a few hundred of them to fill 256 CUs
    return pixels >= CONV3X3_MIN_PIXELS or ((pixels + 255) // 256) * ((cout + 127) // 128) >= CONV3X3_MIN_WGS


def conv3x3_supported(x, conv):
    """``conv3x3_ok`` for an ``nn.Conv2d``: 3x3 kernel, stride 1, square zero padding 0..2, no dilation / groups."""
    return (tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1) and tuple(conv.dilation) == (1, 1) and conv.groups == 1
            and conv.padding in ((0, 0), (1, 1), (2, 2)) and conv.padding_mode == "zeros"
            and conv3x3_ok(x, conv.in_channels, conv.out_channels, conv.padding[0]))


def conv3x3(x, weight, padding):
    """K11: ``conv2d(x, weight, None, stride=1, padding=padding)`` for a 3x3 filter on the bf16 matrix pipe (fp32 products from
    three-way bf16 splits), forward and input gradient (src/nets/base_cnns.py:32-45's hidden convolutions)."""
    return _Conv3x3.apply(x, weight, int(padding))


LINEAR_MIN_ROWS = 32768


def linear_ok(x, in_features, out_features):
    """K11's row-major sibling takes ``x (M, in) @ w (out, in).T`` on a CUDA fp32 contiguous matrix when the inner dimension is a
    multiple of 16, the output width a multiple of 32, and there are enough rows to fill the chip (a workgroup takes 256)."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == in_features and in_features % 16 == 0
            and out_features % 32 == 0 and x.shape[0] >= LINEAR_MIN_ROWS)


def linear_nobias(x, w, mode=0):
    """mode 0: ``x @ w.T``; mode 1: ``x @ w`` (the gradient of mode 0 with respect to its input, x being dY) -- fp32 products formed
    from three-way bf16 splits on the MFMA pipe (csrc/conv.hip::k_linear)."""
    lib = _lib_or_raise()
    x, w = x.contiguous(), w.contiguous()
    M = x.shape[0]
    Nw, Kw = w.shape
    y = torch.empty((M, Nw if mode == 0 else Kw), dtype=torch.float32, device=x.device)
    ws = _workspace("conv", lib.aurppo_conv3x3_wop_bytes(Kw if mode == 0 else Nw, Nw if mode == 0 else Kw), x.device)
    _check(lib.aurppo_linear_f32(_ptr(x), _ptr(w), _ptr(y), M, Kw, Nw, int(mode), C.c_void_p(ws.data_ptr()), _stream()),
           "aurppo_linear_f32")
    return y


def linear_wgrad(gy, x):
    """``gy.T @ x`` -- nn.Linear's weight gradient -- on the bf16 matrix pipe (csrc/conv.hip::k_linear_wgrad: both operands split
    once per workgroup through LDS, the rows cut into slices that are summed in slice order)."""
    lib = _lib_or_raise()
    gy, x = gy.contiguous(), x.contiguous()
    M, N = gy.shape
    K = x.shape[1]
    ws = _workspace("wgrad", lib.aurppo_linear_wgrad_ws_bytes(M, N, K), x.device)
    dw = torch.empty((N, K), dtype=torch.float32, device=x.device)
    _check(lib.aurppo_linear_wgrad_f32(_ptr(gy), _ptr(x.detach()), _ptr(dw), M, N, K, C.c_void_p(ws.data_ptr()), _stream()),
           "aurppo_linear_wgrad_f32")
    return dw


def linear_wgrad_ok(gy, x):
    return (gy.is_cuda and gy.dtype == torch.float32 and gy.dim() == 2 and x.dim() == 2 and gy.shape[1] % 4 == 0 and x.shape[1] % 4 == 0
            and gy.shape[0] >= LINEAR_MIN_ROWS and gy.shape[1] >= 64 and x.shape[1] >= 64)


def linear_bias_act(x, w, bias, act=0):
    """``act(x @ w.T + bias)`` in one kernel (act: 0 none, 1 tanh): csrc/conv.hip::k_linear with the layer's tail in its epilogue."""
    lib = _lib_or_raise()
    x, w = x.contiguous(), w.contiguous()
    M = x.shape[0]
    Nw, Kw = w.shape
    y = torch.empty((M, Nw), dtype=torch.float32, device=x.device)
    ws = _workspace("conv", lib.aurppo_conv3x3_wop_bytes(Kw, Nw), x.device)
    _check(lib.aurppo_linear_bias_act_f32(_ptr(x), _ptr(w), _ptr(bias.contiguous()) if bias is not None else None, _ptr(y), M, Kw, Nw,
                                          int(act), C.c_void_p(ws.data_ptr()), _stream()), "aurppo_linear_bias_act_f32")
    return y


def first_block(obs, state, weight, bias):
    """K10: ``max_pool2d(relu(conv2d(cat[obs, state tiled to a plane], weight, bias, padding=1)), 2)`` -- the first block of
    src/nets/base_cnns.py:28-31 on the input of src/models/robot_actor_critic.py:58-59 -- forward and backward without the
    full-resolution tensors.  ``weight``: (Co, Ci + 1, 3, 3), state plane last; Ci in 1..3, Co a multiple of 16."""
    return _FirstBlock.apply(obs, state, weight, bias)
