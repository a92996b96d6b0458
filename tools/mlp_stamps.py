"""Diagnostic: build the MLP kernels with -DAURPPO_MLP_STAMPS into a SEPARATE library, run one fused step at the
BASELINE minibatch size and print cycle shares per phase (median over workgroups).
Wave 0 of each tile set of k_mlp_step2 (default) or k_mlp_step3 (AURPPO_K7_VARIANT=3): work and barrier-wait cycles per phase."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
so = "/tmp/libaurppo_stamps.so"
csrc = os.path.join(ROOT, "aur_ppo_amd", "csrc")
import __graft_entry__ as g
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                "-DAURPPO_MLP_STAMPS"] + os.environ.get("AURPPO_EXTRA_DEFS", "").split() + [os.path.join(csrc, f) for f in g.HIP_SOURCES] + ["-o", so], check=True)
from aur_ppo_amd import _lib, hip_ops as H
_lib.LIB_PATH = so
_lib._lib = None
from tests.test_mlp_fused import _setup
Hh, pol, bucket, obs, act, rec = _setup(128, 4096, 64, 6)
lay = Hh.mlp_layout(pol, bucket)
M = int(os.environ.get("K7_M", 131072))
idx = torch.randperm(obs.shape[0], device="cuda")[:M].int()
lib = _lib.load()
n = lay["n_params"]
rec64 = Hh.pack_records(rec, act)        # the trainer's layout: record + action row in one 64-B line
for _ in range(3):
    Hh.mlp_ppo_step(obs, None, rec64, idx, bucket.flat_param, lay, bucket.flat_grad, 0.2, 0.0, 0.5)
torch.cuda.synchronize()
ws = H._ws_cache[("mlp", torch.cuda.current_device())]
off = ((8 * (2 * 256 + 8 * 256) + 4 * 256 * n + 63) // 64) * 64
if True:
    if os.environ.get("AURPPO_K7_VARIANT", "3") != "2":
        x = ws[off + 8 * 40 * 256:off + 8 * 44 * 256].view(torch.int64).view(256, 4).cpu().numpy().astype(np.float64)
        x = x[x[:, 0] > 0]
        m = np.median(x, axis=0)
        print(f"k_mlp_step3 set 0, cycles since the phase began (sum over tiles): F1 after the chain {m[0]:.0f}, after issuing W2's loads {m[1]:.0f}; "
              f"B3 after the chain {m[2]:.0f}, after issuing W1's loads {m[3]:.0f}")
    raw = ws[off:off + 8 * 40 * 256].view(torch.int64).view(256, 40).cpu().numpy().astype(np.float64)
    raw = raw[(raw[:, 1] > 0) & (raw[:, 35] > 0) & (raw[:, 35] < 1e6) & (raw[:, 36] > 0) & (raw[:, 36] < 1e7)]   # workgroups of this launch
    cyc, ticks = np.median(raw[:, 32]), np.median(raw[:, 33])
    print(f"shader clock over the tile loop + hand-over: {cyc:.0f} cycles in {ticks:.0f} ticks of the 100 MHz wall clock = {cyc / ticks * 0.1:.2f} GHz")
    ent = raw[:, 34]
    print(f"launch timeline (10-ns ticks of the wall clock, over {raw.shape[0]} workgroups): first-to-last workgroup entry "
          f"{(ent.max() - ent.min()) / 100:.1f} us; medians: prologue {np.median(raw[:, 35]) / 100:.1f} us, tile loop "
          f"{np.median(raw[:, 36]) / 100:.1f} us (min {raw[:, 36].min() / 100:.1f}, max {raw[:, 36].max() / 100:.1f}), wait for the "
          f"other set {np.median(raw[:, 37]) / 100:.1f} us, hand-over + slab {np.median(raw[:, 38]) / 100:.1f} us; "
          f"last exit - first entry {((ent + raw[:, 35:39].sum(1)).max() - ent.min()) / 100:.1f} us")
    st = raw[:, :32].reshape(-1, 2, 16)
    names = ["S  land tile, prefetch", "F1 layer 1 + tanh", "F2 layer 2 + tanh", "F3 head", "L  loss lanes", "B1 dH2,dW3,dZ2",
             "B2 dW2,dH1,dZ1", "B3 dW1"]
    med = np.median(st, axis=0)     # (2, 16)
    for s_ in range(2):
        tot = med[s_].sum()
        print(f"-- set {s_}: total {tot:.0f} cycles over the tile loop ({st.shape[0]} workgroups with work)")
        for k, nm in enumerate(names):
            print(f"   {nm:24s} work {med[s_, k]:10.0f} ({100 * med[s_, k] / tot:5.1f} %)   barrier wait {med[s_, 8 + k]:10.0f} ({100 * med[s_, 8 + k] / tot:5.1f} %)")
