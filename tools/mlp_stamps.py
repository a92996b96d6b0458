"""Diagnostic: build mlp.hip with -DAURPPO_MLP_STAMPS into a SEPARATE library, run one fused step at the
BASELINE minibatch size and print wave 0's cycle share per phase (median over workgroups)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
so = "/tmp/libaurppo_stamps.so"
csrc = os.path.join(ROOT, "aur_ppo_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                "-DAURPPO_MLP_STAMPS"] + [os.path.join(csrc, f) for f in ("gae.hip", "shuffle.hip", "gather.hip", "loss.hip", "clip.hip", "mlp.hip", "api.hip")] + ["-o", so], check=True)
from aur_ppo_amd import _lib, hip_ops as H
_lib.LIB_PATH = so
_lib._lib = None
from tests.test_mlp_fused import _setup
Hh, pol, bucket, obs, act, rec = _setup(128, 4096, 64, 6)
lay = Hh.mlp_layout(pol, bucket)
M = 131072
idx = torch.randperm(obs.shape[0], device="cuda")[:M].int()
lib = _lib.load()
n = lay["n_params"]
ws_bytes = lib.aurppo_mlp_workspace_bytes(n)
for _ in range(3):
    Hh.mlp_ppo_step(obs, act, rec, idx, bucket.flat_param, lay, bucket.flat_grad, 0.2, 0.0, 0.5)
torch.cuda.synchronize()
ws = H._ws_cache[("mlp", torch.cuda.current_device())]
off = ((8 * (2 * 256 + 8 * 256) + 4 * 256 * n + 63) // 64) * 64
st = ws[off:off + 8 * 16 * 256].view(torch.int64).view(256, 16).cpu().numpy().astype(np.float64)
names = ["land X/act->LDS+bar", "issue prefetch", "L1 mma+tanh+bar", "L2 mma+tanh+bar", "head mma+bar", "loss lanes+bar",
         "dH2,dZ2,dW3+bar", "dW2,dH1,dZ1+bar", "dZ1->LDS+bar", "dW1+bar", "(pre-slab)", "slab+reduce tail"]
med = np.median(st, axis=0)
tot = med[:12].sum()
for k, nm in enumerate(names):
    print(f"{nm:24s} {med[k]:12.0f} cycles  {100 * med[k] / tot:5.1f} %")
print("total cycles (wave 0, median WG):", tot, " tiles per WG:", M // 32 // 256)
