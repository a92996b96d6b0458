"""k_fy_accept3 (AURPPO_K2_ACCEPT=3) against numpy over sizes from one chunk to many, several relay widths, consecutive shuffles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AURPPO_TEST_KNOBS"] = "1"
os.environ["AURPPO_K2_ACCEPT"] = "3"
import numpy as np, torch
from aur_ppo_amd import hip_ops as H
for G in (1, 2, 6, 8):
    os.environ["AURPPO_K2_ACCEPT3_WGS"] = str(G)
    for B in (2, 10, 1000, 20000, 40000, 70000, 300000, 524288):
        rng = H.MT19937(7, B, torch.device("cuda"))
        out = torch.empty((3, B), dtype=torch.int32, device="cuda")
        got = rng.shuffle_epochs(B, 3, out=out).cpu().numpy()
        st = np.random.RandomState(7)
        idx = np.arange(B)
        ok = True
        for e in range(3):
            st.shuffle(idx)
            ok = ok and np.array_equal(got[e], idx)
        print(f"G={G} B={B}: bit-exact {ok}, status {float(rng.status()) if hasattr(rng, 'status') else 'n/a'}", flush=True)
        assert ok
