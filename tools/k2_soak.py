"""Soak: K2_SOAK_N consecutive shuffles of 524 288 on one handle (the stream carried across all of them), every permutation
compared with numpy's -- the rare paths of the accept relay over a long stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aur_ppo_amd import hip_ops as H
n, total, E = 524288, int(os.environ.get("K2_SOAK_N", "1200")), 8
rng = H.MT19937(11, n)
rs = np.random.RandomState(11)
out = torch.empty((E, n), dtype=torch.int32, device="cuda")
t0 = time.time()
for k in range(total // E):
    got = rng.shuffle_epochs(n, E, out=out).cpu().numpy()
    ref = np.arange(n)            # shuffle_epochs starts every call from arange and keeps shuffling it (src/ppo.py:213-217)
    for e in range(E):
        rs.shuffle(ref)
        assert np.array_equal(got[e], ref), (k, e)
    if k % 25 == 0:
        print(f"{(k + 1) * E} shuffles ok ({time.time() - t0:.0f} s)", flush=True)
status = torch.zeros(1, device="cuda")
rng.status_into(status)
assert float(status) == 0.0
print(f"{total // E * E} shuffles ok, status clean")
