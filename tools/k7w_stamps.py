"""Diagnostic: k_mlpw3_step built with -DK7W_STAMPS into a SEPARATE library; one fused step of a wide policy at the BASELINE
minibatch size; prints thread 0's cycles per segment of the tile loop (work up to the closing barrier / wait inside it), medians
over the workgroups.   python tools/k7w_stamps.py [hidden=128] [layers=3]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
so = "/tmp/libaurppo_k7w_stamps.so"
csrc = os.path.join(ROOT, "aur_ppo_amd", "csrc")
import __graft_entry__ as g
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                "-DK7W_STAMPS"] + os.environ.get("AURPPO_EXTRA_DEFS", "").split() + [os.path.join(csrc, f) for f in g.HIP_SOURCES] + ["-o", so], check=True)
from aur_ppo_amd import _lib, hip_ops as H
_lib.LIB_PATH = so
_lib._lib = None
hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 128
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 3
from tests.test_mlp_wide import _setup
Hh, pol, bucket, obs, act, rec = _setup(128, 4096, 64, 6, hidden, layers)
lay = Hh.mlp_layout(pol, bucket)
M = int(os.environ.get("K7_M", 131072))
idx = torch.randperm(obs.shape[0], device="cuda")[:M].int()
rec64 = Hh.pack_records(rec, act)
for _ in range(3):
    Hh.mlp_ppo_step(obs, None, rec64, idx, bucket.flat_param, lay, bucket.flat_grad, 0.2, 0.0, 0.5)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((512, 32), dtype=np.uint64)
assert lib.aurppo_k7w_stamps_read(buf.ctypes.data_as(C.c_void_p)) == 0
x = buf.astype(np.float64)
x = x[x[:, 21] > 0]
tiles = np.median(x[:, 21])
names = ["land tile + prefetch", "F1", "F2", "F3", "head", "loss lanes", "head bwd (dH, dW3, dZ)", "B (l=1: dW2, dH1)", "B (l=2: dW3.., dH2)", "dW1"]
med = np.median(x, axis=0)
tot = med[:20].sum()
print(f"{x.shape[0]} workgroups, {tiles:.0f} tiles each (median); prologue {med[20]:.0f} cycles; entry -> end of loop {med[22]:.0f}; per tile {tot / tiles:.0f} cycles")
for k, nm in enumerate(names):
    print(f"  {nm:28s} work {med[k] / tiles:8.0f}   barrier wait {med[10 + k] / tiles:8.0f}   (per tile)")
print(f"  inside the forward phases (per tile): F1 chain {med[24] / tiles:.0f}, row / index requests {med[25] / tiles:.0f}, F1 next slice + tanh + stores {med[26] / tiles:.0f}; "
      f"F2.. chains (sum) {med[28] / tiles:.0f}, next slice + tanh + stores (sum) {med[29] / tiles:.0f}")
