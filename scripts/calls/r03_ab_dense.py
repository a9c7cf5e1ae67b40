"""A/B timing of the dense step's kernels on ONE box: Gram strip kernel (HIOPAMD_GRAM_LDS) and GEMV stage 1 (HIOPAMD_GEMV) variants are
chosen per PROCESS (the switches are read once), so this script is run once per setting by scripts/calls/r03_gpu_13.sh."""
import os, sys, time
import torch
sys.path.insert(0, ".")
from hiop_amd.runtime import Context
ctx = Context(0)
out = []
for k, n in ((200, 1_250_000), (100, 1_000_000)):
    l = 6
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    X = torch.rand(k, n, generator=g, device="cuda", dtype=torch.float64)
    S = torch.rand(l, n, generator=g, device="cuda", dtype=torch.float64)
    Y = torch.rand(l, n, generator=g, device="cuda", dtype=torch.float64)
    d = torch.rand(n, generator=g, device="cuda", dtype=torch.float64)
    x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64)
    kw = k + 2 * l
    W = torch.zeros(k, kw, device="cuda", dtype=torch.float64)
    y = torch.zeros(k, device="cuda", dtype=torch.float64)
    torch.cuda.synchronize()
    def gram(): ctx.call("hiopamd_gram_weighted_stacked", k, n, X, n, k, X, n, l, S, n, l, Y, n, d, 0.0, W, kw, 1.0)
    def gemv(): ctx.call("hiopamd_mat_times_vec", k, n, X, n, 0.0, y, 1.0, x)
    for name, fn in (("gram", gram), ("gemv_n", gemv)):
        for _ in range(3): fn()
        ctx.sync()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(10): fn()
            ctx.sync()
            best = min(best, (time.perf_counter() - t0) / 10)
        out.append(f"{name} k={k}: {best*1e3:.3f} ms")
    del X, S, Y, d, x
print(" | ".join(out))
