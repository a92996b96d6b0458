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
