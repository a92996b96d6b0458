"""How do the library GEMMs do on the policy's skinny shapes?  (131072 x 64) @ (64 x 64) etc."""
import os, sys, torch, json
import torch.nn.functional as F
dev = torch.device("cuda")
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return b.elapsed_time(e) * 1e3 / iters
print("prefer hipblaslt env:", os.environ.get("TORCH_BLAS_PREFER_HIPBLASLT"), "backend:", torch.backends.cuda.preferred_blas_library())
for M in (131072, 16384):
    x = torch.randn(M, 64, device=dev); h = torch.randn(M, 64, device=dev)
    for (K, Nn) in ((64, 64), (64, 6), (64, 1), (64, 128)):
        W = torch.randn(Nn, K, device=dev); b = torch.randn(Nn, device=dev)
        t_lin = timeit(lambda: F.linear(x, W, b))
        t_mm = timeit(lambda: x @ W.t())
        go = torch.randn(M, Nn, device=dev)
        t_dx = timeit(lambda: go @ W)             # grad input
        t_dw = timeit(lambda: go.t() @ x)         # grad weight
        fl = 2 * M * K * Nn
        print(f"M={M} K={K} N={Nn}: linear {t_lin:8.1f} us ({fl/t_lin/1e6:7.2f} TF)  mm {t_mm:8.1f}  dX {t_dx:8.1f}  dW {t_dw:8.1f}")
    t = timeit(lambda: torch.tanh(h))
    print(f"M={M} tanh(64 wide): {t:.1f} us")
