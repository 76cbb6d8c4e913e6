"""tools/rdbw.py — what a plain streaming read of the benchmark buffer reaches on this box (torch reductions),
the practical ceiling the HBM-bound kernels are compared with in DESIGN.md."""
import time
import torch
n = 8192 * 480000 * 2
x = torch.empty(n, dtype=torch.float32, device="cuda").uniform_(-1, 1)
for name, fn in (("sum", lambda: x.sum()), ("amax", lambda: x.amax()), ("abs().amax (2 passes? fused)", lambda: x.abs().amax())):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {dt * 1e3:.2f} ms  {n * 4 / dt / 1e12:.2f} TB/s")
