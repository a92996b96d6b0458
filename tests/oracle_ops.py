"""Stand-in for ``aur_ppo_amd.hip_ops`` built on the CPU oracle -- TEST-ONLY.  It lets the host
logic of the trainer (minibatch slicing, LR anneal, logging, early stop, env sharding, gradient
all-reduce) run on a box without a GPU.  The product never imports this; its default ``ops`` is
the HIP module, which raises without the built library or without a GPU."""
import numpy as np
import torch

from oracle import c_oracle as CO
from oracle import ppo_oracle as O

GAE, NORMAL_ADV, GAE_SKIP_LAST = 0, 1, 2
VLOSS_RETURNS, VLOSS_CLIPPED, VLOSS_OLDVALUES = 0, 1, 2
S_LOSS, S_PG, S_VL, S_ENT, S_OLD_KL, S_KL, S_CLIPFRAC, S_ADV_MEAN, S_ADV_STD = range(9)
N_SCALARS = 9


def gae(rewards, values, terminals, next_value, next_done, gamma, lam, mode=GAE, out=None, log_probs=None, rec=None):
    ret, adv = O.gae(rewards.numpy(), values.numpy(), terminals.numpy(), next_value.numpy(), next_done.numpy(),
                     gamma, lam, mode)
    ret, adv = torch.from_numpy(ret), torch.from_numpy(adv)
    if rec is not None:
        rec.copy_(torch.stack([log_probs.reshape(-1), adv.reshape(-1), ret.reshape(-1), values.reshape(-1)], 1))
    return ret, adv


class MT19937:
    def __init__(self, seed, max_n, device=None):
        self.rs = np.random.RandomState(seed)
        self.max_n = max_n

    def seed(self, seed):
        self.rs = np.random.RandomState(seed)

    def get_state(self):
        st = self.rs.get_state()
        return st[1].copy(), int(st[2])

    def set_state(self, key, pos):
        self.rs.set_state(("MT19937", np.asarray(key, dtype=np.uint32), int(pos), 0, 0.0))

    def status_into(self, out):
        out.zero_()
        return out

    def shuffle_(self, idx):
        x = idx.numpy()
        self.rs.shuffle(x)
        return idx

    def shuffle_epochs(self, n, epochs, out=None):
        perms = O.epoch_permutations(self.rs, n, epochs)
        t = torch.from_numpy(np.stack(perms).astype(np.int32))
        if out is not None:
            out.copy_(t)
            return out
        return t


def gather(idx, srcs, outs=None):
    return [s[idx.long()].contiguous() for s in srcs]


class _Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, newlogp, newv, entropy, oldlogp, adv, oldv, ret, clip, ent_coef, vf_coef, norm_adv, vloss_mode,
                out_scalars):
        n = lambda t: t.detach().reshape(-1).numpy()
        sc, g_lp, g_v, g_e = CO.ppo_loss(n(newlogp), n(oldlogp), n(adv), n(newv), n(oldv), n(ret), n(entropy), clip,
                                         ent_coef, vf_coef, norm_adv, vloss_mode)
        if out_scalars is not None:
            out_scalars.copy_(torch.from_numpy(sc))
        ctx.save_for_backward(torch.from_numpy(g_lp), torch.from_numpy(g_v), torch.from_numpy(g_e))
        ctx.v_shape = newv.shape
        return torch.tensor(sc[S_LOSS])

    @staticmethod
    def backward(ctx, grad_out):
        g_lp, g_v, g_e = ctx.saved_tensors
        return (g_lp * grad_out, (g_v * grad_out).view(ctx.v_shape), g_e * grad_out) + (None,) * 10


def ppo_loss(newlogp, newv, entropy, oldlogp, adv, oldv, ret, clip, ent_coef, vf_coef, norm_adv=True,
             vloss_mode=VLOSS_CLIPPED, out_scalars=None):
    return _Loss.apply(newlogp, newv, entropy, oldlogp, adv, oldv, ret, clip, ent_coef, vf_coef, norm_adv, vloss_mode,
                       out_scalars)


def ppo_loss_packed(newlogp, newv, entropy, rec, clip, ent_coef, vf_coef, norm_adv=True, vloss_mode=VLOSS_CLIPPED,
                    out_scalars=None):
    return _Loss.apply(newlogp, newv, entropy, rec[:, 0], rec[:, 1], rec[:, 3], rec[:, 2], clip, ent_coef, vf_coef,
                       norm_adv, vloss_mode, out_scalars)


def grad_norm_clip_(flat_grads, max_norm, out_norm=None):
    g, norm = CO.grad_norm_clip(flat_grads.numpy(), max_norm)
    flat_grads.copy_(torch.from_numpy(g))
    if out_norm is not None:
        out_norm.fill_(norm)
        return out_norm
    return torch.tensor([norm])


def loss_fwd_bwd(newlogp, oldlogp, adv, newv, oldv, ret, entropy, clip, ent_coef, vf_coef, norm_adv=True,
                 vloss_mode=VLOSS_CLIPPED, out_scalars=None):
    n = lambda t: t.detach().reshape(-1).numpy()
    sc, g_lp, g_v, g_e = CO.ppo_loss(n(newlogp), n(oldlogp), n(adv), n(newv), n(oldv), n(ret), n(entropy), clip, ent_coef,
                                     vf_coef, norm_adv, vloss_mode)
    return torch.from_numpy(sc), torch.from_numpy(g_lp), torch.from_numpy(g_v), torch.from_numpy(g_e)
