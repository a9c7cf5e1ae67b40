"""Time the stacked Gram of the dense path alone (k = 200, l = 6, n = 1.25e6), no correctness check (debug variants)."""
import ctypes as C, sys, time, torch
sys.path.insert(0, ".")
from hiop_amd.runtime import Context
ctx = Context(0)
k, l, n = 200, 6, 1_250_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.rand(k, n, generator=g, device="cuda", dtype=torch.float64)
S = torch.rand(l, n, generator=g, device="cuda", dtype=torch.float64)
Y = torch.rand(l, n, generator=g, device="cuda", dtype=torch.float64)
d = torch.rand(n, generator=g, device="cuda", dtype=torch.float64)
kw = k + 2 * l
W = torch.zeros(k, kw, device="cuda", dtype=torch.float64)
torch.cuda.synchronize()
def run():
    ctx.call("hiopamd_gram_weighted_stacked", k, n, X, n, k, X, n, l, S, n, l, Y, n, d, 0.0, W, kw, 1.0)
for _ in range(3): run()
ctx.sync()
t0 = time.perf_counter()
for _ in range(10): run()
ctx.sync()
dt = (time.perf_counter() - t0) / 10
print(f"stacked gram {dt*1e3:.3f} ms  useful {2.0*k*kw*n/dt/1e12:.1f} TFLOP/s")
