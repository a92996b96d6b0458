"""Diagnostic: build the library with -DAURPPO_ACC_STAMPS into /tmp, run shuffles of 524288 and print where
thread 0 of k_fy_accept spends its cycles per step."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
so = "/tmp/libaurppo_accstamps.so"
csrc = os.path.join(ROOT, "aur_ppo_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                "-DAURPPO_ACC_STAMPS"] + os.environ.get("AURPPO_EXTRA_DEFS", "").split() + [os.path.join(csrc, f) for f in g.HIP_SOURCES] + ["-o", so], check=True)
from aur_ppo_amd import _lib, hip_ops as H
_lib.LIB_PATH = so
_lib._lib = None
n = 524288
rng = H.MT19937(1, n)
out = torch.empty((4, n), dtype=torch.int32, device="cuda")
import time
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rng.shuffle_epochs(n, 4, out=out)
    torch.cuda.synchronize()
    t_ms = (time.perf_counter() - t0) * 1e3
print(f"4 shuffles of {n}: {t_ms:.3f} ms (stamped build)")
lib = _lib.load()
buf = (C.c_longlong * 24)()
lib.aurppo_debug_accept_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
assert lib.aurppo_debug_accept_stamps(rng._h, buf) == 0
v = list(buf)
if os.environ.get("AURPPO_K2_ACCEPT", "3") != "1":
    names3 = ["draws + next fetch + reference", "solve from the guess", "list the sensitive draws", "wait for the predecessor", "walk the list + publish",
              "redo counts from the true start", "emit targets"]
    ch = max(v[7], 1)
    tot = sum(v[:7])
    print(f"k_fy_accept3, thread 0 of workgroup 1 (last shuffle): {v[7]} chunks, fast path {v[8]}, list usable {v[9]}, entries per chunk {v[10] / ch:.0f}, "
          f"mean |true - guessed start| {v[11] / ch:.1f}, W {v[13] / ch:.0f}, rounds {v[12]} ({v[12] / ch:.1f} per chunk); {tot} cycles")
    for k, nm in enumerate(names3):
        print(f"  {nm:34s} {v[k]:10d} cycles  {100 * v[k] / tot:5.1f} %   {v[k] / ch:8.0f} per chunk")
    sys.exit(0)
names = ["wait draws", "issue next fetch", "first guess run", "rounds: combine + recount", "emit run", "bookkeeping", "rounds: scan + publish", "rounds: barrier"]
steps, iters = v[8], v[9]
tot = sum(v[:8])
print(f"steps {steps}, fixed-point iterations {iters} ({iters / max(steps, 1):.2f} per step), total {tot} cycles")
for k, nm in enumerate(names):
    print(f"  {nm:20s} {v[k]:10d} cycles  {100 * v[k] / tot:5.1f} %   {v[k] / max(steps, 1):8.0f} per step")
