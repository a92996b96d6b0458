"""Stand-alone time of the E = 4 epoch shuffles of one update (B = 524 288) and bit-exactness against numpy;
AURPPO_K2_ACCEPT=1|2|3 selects the accept kernel (default 3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from aur_ppo_amd import _lib, hip_ops as H
if os.environ.get("AURPPO_LIB"):      # another build of the library (timing experiments)
    _lib.LIB_PATH = os.environ["AURPPO_LIB"]
    _lib._lib = None
B, E = int(os.environ.get("K2_B", 524288)), 4
rng = H.MT19937(1, B, torch.device("cuda"))
out = torch.empty((E, B), dtype=torch.int32, device="cuda")
got = rng.shuffle_epochs(B, E, out=out).cpu().numpy()
st = np.random.RandomState(1)
idx = np.arange(B)
ok = True
for e in range(E):
    st.shuffle(idx)
    ok = ok and np.array_equal(got[e], idx)
ts = []
for _ in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rng.shuffle_epochs(B, E, out=out)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"accept kernel {os.environ.get('AURPPO_K2_ACCEPT', 'default')}: bit-exact vs numpy: {ok}; {E} shuffles of {B}: median {np.median(ts[2:]):.3f} ms (min {min(ts[2:]):.3f})")
