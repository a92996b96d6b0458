"""Time the K7 kernel alone (hipEvents around the main kernel) at the headline minibatch for the variant in AURPPO_K7_VARIANT;
AURPPO_LIB=<path> loads another build of the library (timing experiments)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from aur_ppo_amd import _lib
if os.environ.get("AURPPO_LIB"):
    _lib.LIB_PATH = os.environ["AURPPO_LIB"]
from aur_ppo_amd import hip_ops as H
from tests.test_mlp_fused import _setup
M = int(os.environ.get("K7_M", 131072))
Hh, pol, bucket, obs, act, rec = _setup(128, 4096, 64, 6)
lay = Hh.mlp_layout(pol, bucket)
idx = torch.randperm(obs.shape[0], device="cuda")[:M].int()
rec64 = Hh.pack_records(rec, act)
ts = []
for it in range(30):
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    Hh.mlp_ppo_step(obs, None, rec64, idx, bucket.flat_param, lay, bucket.flat_grad, 0.2, 0.0, 0.5, events=ev)
    torch.cuda.synchronize()
    if it >= 5:
        ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
print(f"variant {os.environ.get('AURPPO_K7_VARIANT', 'default')} M={M}: main kernel {np.median(ts):.1f} us (min {min(ts):.1f}, max {max(ts):.1f})")
