"""Calibration launches for FETCH_SIZE on this access pattern (run under rocprofv3 --pmc FETCH_SIZE):
a plain copy of the obs bytes, the gather with identity indices (streaming), the gather with a
random permutation -- each on buffers larger than the Infinity Cache rotation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aur_ppo_amd import hip_ops as H
dev = torch.device("cuda")
B, M, D = 524288, 131072, 64
srcs = [torch.randn(B, D, device=dev) for _ in range(3)]
out = torch.empty(M, D, device=dev)
ident = torch.arange(M, device=dev, dtype=torch.int32)
perm = H.MT19937(1, B).shuffle_epochs(B, 1)[0][:M].contiguous()
flush = torch.empty(80 * 1024 * 1024, device=dev)      # 320 MB
for rep in range(3):
    for s in srcs:
        flush.zero_()                                   # evict
        out.copy_(s[:M])                                # elementwise copy kernel (float4 streaming)
        flush.zero_()
        H.gather(ident, [s], [out])                     # k_gather<1,1>, sequential rows
        flush.zero_()
        H.gather(perm, [s], [out])                      # k_gather<1,1>, random rows
torch.cuda.synchronize()
print("done")
