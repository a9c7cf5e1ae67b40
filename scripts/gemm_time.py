"""timesMat on the matrix pipe against the scalar LDS kernel (HIOPAMD_GEMM=0), 2000 x 2000 x 2000, HIP events on the context's stream
(scripts/calls/r04_gpu_16.sh)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hiop_amd.runtime import Context
ctx = Context(0)
n = 2000
A = torch.rand(n, n, device="cuda", dtype=torch.float64); X = torch.rand(n, n, device="cuda", dtype=torch.float64); W = torch.zeros(n, n, device="cuda", dtype=torch.float64)
L = ctx._L
pa, px, pw = C.c_void_p(A.data_ptr()), C.c_void_p(X.data_ptr()), C.c_void_p(W.data_ptr())
torch.cuda.synchronize()
ms = bench.time_on_ctx_stream(ctx, lambda: L.hiopamd_mat_times_mat(ctx.h, n, n, n, pa, n, 0.0, pw, n, 1.0, px, n), reps=10)
ctx.sync()
err = float((W - A @ X).abs().max())
print("timesMat %d^3: %.3f ms = %.1f TFLOP/s, max error %.2e (HIOPAMD_GEMM=%s)" % (n, ms, 2 * n ** 3 / ms / 1e9, err, os.environ.get("HIOPAMD_GEMM", "mfma")))
