"""One flat fp32 bucket for all parameters and one for all gradients.

The reference updates ~17 k parameters spread over 13 small tensors (src/ppo.py:80,266-269).
On MI355X every per-tensor kernel is launch-bound, and the multi-GPU exchange wants ONE message
(68 KB for the MLP: latency-bound on xGMI), so parameters and gradients are re-homed as views
into two contiguous buffers: the RCCL all-reduce, the K6 norm-clip kernel and Adam each touch a
single tensor."""
from __future__ import annotations

import torch


class FlatBucket:
    def __init__(self, params):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FlatBucket: no parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        n = sum(p.numel() for p in self.params)
        pad = (-n) % 4                      # keep the bucket float4-sized
        self.numel = n
        self.flat_param = torch.zeros(n + pad, device=dev, dtype=dt)
        self.flat_grad = torch.zeros(n + pad, device=dev, dtype=dt)
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                view = self.flat_param[off:off + k].view_as(p)
                view.copy_(p)
                p.data = view
                p.grad = self.flat_grad[off:off + k].view_as(p)
                off += k

    def zero_grad(self):
        """One memset instead of per-tensor zeroing; .grad views stay attached (autograd then
        accumulates in place)."""
        self.flat_grad.zero_()

    def check_attached(self):
        lo = self.flat_grad.data_ptr()
        hi = lo + self.flat_grad.numel() * 4
        for p in self.params:
            if p.grad is None or not (lo <= p.grad.data_ptr() < hi):
                raise RuntimeError("a parameter's .grad was detached from the flat bucket "
                                   "(use FlatBucket.zero_grad(), not optimizer.zero_grad())")


class FlatAdamMixin:
    """clip_grad_norm_ + Adam on the flat bucket, shared by ``ppo`` and ``robot_ppo`` (src/ppo.py:80,266-269;
    src/robot_ppo.py:401-402).  Expects ``self.ops``, ``self.device``, ``self.bucket``, ``self.optimizer``,
    ``self._lr_tensor`` (device scalar or None) and ``self.max_grad_norm``."""

    def _adam_setup(self):
        """Flat Adam state for K6b (clip + Adam fused).  The torch optimizer object stays -- upstream code
        reads ``optimizer.param_groups`` / ``state_dict()`` -- with its per-parameter state entries aliased to
        views of the flat moment buffers and a shared device step counter."""
        self._fused_adam = self.device.type == "cuda" and hasattr(self.ops, "clip_adam_")
        if not self._fused_adam:
            return
        fp = self.bucket.flat_param
        self._adam_m, self._adam_v = torch.zeros_like(fp), torch.zeros_like(fp)
        self._adam_t = torch.zeros(1, device=self.device)
        off = 0
        for p in self.bucket.params:
            k = p.numel()
            self.optimizer.state[p] = {"step": self._adam_t, "exp_avg": self._adam_m[off:off + k].view_as(p),
                                       "exp_avg_sq": self._adam_v[off:off + k].view_as(p)}
            off += k

    def _clip_and_step(self, norm_out, clip_n=None):
        """clip_grad_norm_ over the first ``clip_n`` elements (all by default) + optimizer.step()."""
        if self._fused_adam:
            g = self.optimizer.param_groups[0]
            self.ops.clip_adam_(self.bucket.flat_param, self.bucket.flat_grad, self._adam_m, self._adam_v, self._lr_tensor,
                                self._adam_t, self.max_grad_norm, clip_n, g["betas"], g["eps"], norm_out)
        else:
            fg = self.bucket.flat_grad if clip_n is None else self.bucket.flat_grad[:clip_n]
            self.ops.grad_norm_clip_(fg, self.max_grad_norm, norm_out)
            self.optimizer.step()

    def set_lr(self, lr):
        g = self.optimizer.param_groups[0]
        if self._lr_tensor is not None:
            self._lr_tensor.fill_(float(lr))
            g["lr"] = self._lr_tensor
        else:
            g["lr"] = float(lr)

    def _adopt_lr(self):
        """Upstream writes ``optimizer.param_groups[0]["lr"] = lrnow`` (src/ppo.py:198).  If a caller did
        that, move the value into the device scalar the (possibly captured) Adam step reads."""
        g = self.optimizer.param_groups[0]
        if self._lr_tensor is not None and g["lr"] is not self._lr_tensor:
            self._lr_tensor.fill_(float(g["lr"]))
            g["lr"] = self._lr_tensor

    def get_lr(self):
        return float(self.optimizer.param_groups[0]["lr"])

    # ---- resume (absent upstream): everything an interrupted run needs besides the weights
    def trainer_state(self):
        """Flat Adam moments + step count, learning rate and the device shuffle generator's (key, pos) -- numpy's
        ``get_state()`` twin -- as CPU tensors / arrays, for ``torch.save`` next to the reference's ``.pt`` keys."""
        st = {"lr": self.get_lr()}
        if getattr(self, "_fused_adam", False):
            st.update(adam_m=self._adam_m.detach().cpu().clone(), adam_v=self._adam_v.detach().cpu().clone(),
                      adam_t=float(self._adam_t))
        if getattr(self, "rng", None) is not None:
            key, pos = self.rng.get_state()
            st.update(rng_key=torch.from_numpy(key.astype("int64")), rng_pos=int(pos))
            # ``ppo`` draws the NEXT update's permutations ahead of time: the generator already sits behind them, so
            # they belong to the state (the resumed run must step through exactly these, then continue the stream)
            ahead = getattr(self, "_perms", None)
            if ahead is not None:
                ready = getattr(self, "_perm_ready", None)
                if ready is not None:
                    ready.synchronize()
                st["perms_ahead"] = ahead.detach().cpu().clone()
        return st

    def load_trainer_state(self, st):
        if getattr(self, "_fused_adam", False) and "adam_m" in st:
            self._adam_m.copy_(st["adam_m"])
            self._adam_v.copy_(st["adam_v"])
            self._adam_t.fill_(float(st["adam_t"]))
        if "rng_key" in st:
            if getattr(self, "rng", None) is None:
                self.seed_all(1)
            import numpy as np
            self.rng.set_state(st["rng_key"].numpy().astype(np.uint32), int(st["rng_pos"]))
            if hasattr(self, "_perms"):
                # permutations this trainer drew ahead came from its old stream position; the saved run's replace them
                self._perms = st["perms_ahead"].to(self.device) if "perms_ahead" in st else None
                self._perm_ready = None
        self.set_lr(st["lr"])

    def _checkpoint_nets(self):
        """(key, module) pairs saved beside the optimizer: overridden by robot_ppo to write upstream's key layout."""
        return [("policy_state", self.policy)]

    def save_checkpoint(self, path, update=0):
        """Weights + optimizer + ``trainer_state()`` in one ``.pt``.  ``robot_ppo`` keeps upstream's keys
        (``actor_state`` / ``critic_state`` / ``optimizer_state``, src/robot_ppo.py:502-507) and adds to them."""
        sd = {k: {n: v.detach().cpu().clone() for n, v in m.state_dict().items()} for k, m in self._checkpoint_nets()}
        osd = self.optimizer.state_dict()
        # plain numbers in the param groups (the capturable optimizer keeps lr in a device tensor) and CPU state tensors:
        # the file then holds nothing but tensors, numbers, strings, lists and dicts and loads with ``weights_only=True``
        osd["param_groups"] = [{k: (float(v) if torch.is_tensor(v) else v) for k, v in g.items()} for g in osd["param_groups"]]
        osd["state"] = {i: {k: (v.detach().cpu().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                        for i, st in osd["state"].items()}
        sd["optimizer_state"] = osd
        sd["trainer_state"] = self.trainer_state()
        sd["update"] = int(update)
        extra = getattr(self, "_checkpoint_extra", None)
        if extra is not None:
            sd.update(extra())
        torch.save(sd, path)

    def load_checkpoint(self, path):
        """Inverse of ``save_checkpoint``; returns the number of updates the saved run had completed.  Also accepts a
        reference-format file (the weights and ``optimizer_state`` only, src/robot_ppo.py:502-507) and a checkpoint written
        on the other optimizer path: Adam's moments are then taken from ``optimizer_state``."""
        sd = torch.load(path, map_location="cpu", weights_only=True)
        with torch.no_grad():
            for k, m in self._checkpoint_nets():
                tgt = m.state_dict()
                for n, v in sd[k].items():
                    tgt[n].copy_(v)              # in place: parameters stay views of the flat bucket
            restore = getattr(self, "_checkpoint_restore", None)
            if restore is not None:
                restore(sd)
        ts = sd.get("trainer_state") or {}
        osd = sd.get("optimizer_state")
        if not getattr(self, "_fused_adam", False):
            if osd is not None:
                self.optimizer.load_state_dict(osd)
        elif "adam_m" not in ts and osd is not None and osd.get("state"):
            # saved by torch's per-parameter Adam (the CPU path, or upstream itself): re-home exp_avg / exp_avg_sq / step in the
            # flat moment buffers K6b reads.  ``state_dict()`` numbers a state entry by the parameter's position in the
            # optimizer's param groups, which need not be the bucket's order (robot_ppo: actor first, then the rest)
            pos, n_seen = {}, 0
            for grp in self.optimizer.param_groups:
                for q in grp["params"]:
                    pos[id(q)] = n_seen
                    n_seen += 1
            with torch.no_grad():
                off, step = 0, 0.0
                for p in self.bucket.params:
                    k = p.numel()
                    st = osd["state"].get(pos.get(id(p), -1))
                    if st is not None:
                        if st["exp_avg"].numel() != k or st["exp_avg_sq"].numel() != k:
                            raise RuntimeError(f"optimizer_state entry {pos[id(p)]} has {st['exp_avg'].numel()} elements, "
                                               f"the parameter it belongs to has {k}")
                        self._adam_m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                        self._adam_v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                        step = max(step, float(st["step"]))
                    off += k
                self._adam_t.fill_(step)
        if "lr" not in ts:
            ts = dict(ts, lr=(float(osd["param_groups"][0]["lr"]) if osd is not None else self.get_lr()))
        self.load_trainer_state(ts)
        self.bucket.check_attached()
        return int(sd.get("update", 0))
