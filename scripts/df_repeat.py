"""Dataflow LDL^T repeated many times on one solver object (and on fresh objects): any bounded-wait time-out shows up as an
exception with the library's diagnostic record on stderr.  DF_N (8192), DF_REPS (300), DF_OBJECTS (4); DF_VERIFY=1: solve with every
factor and check the residual."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hiop_amd.runtime import Context
from hiop_amd.kkt import LinSolverSymDense
N = int(os.environ.get("DF_N", "8192")); reps = int(os.environ.get("DF_REPS", "300")); nobj = int(os.environ.get("DF_OBJECTS", "4"))
ctx = Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
M = torch.rand(N, N, generator=g, device="cuda", dtype=torch.float64) * 1e-3
M = M + M.T + torch.eye(N, device="cuda", dtype=torch.float64) * 10.0
def kfd_evicted():
    """KFD's per-process accounting of how long this process's queues were EVICTED (preempted, context-saved): ms per GPU, or None"""
    import glob
    out = {}
    # (inside a container os.getpid() is not the pid the driver knows: list every process the driver shows)
    for f in glob.glob("/sys/class/kfd/kfd/proc/*/stats_*/evicted_ms"):
        try:
            v = int(open(f).read().strip())
        except Exception as e:
            v = repr(e)[:40]
        out[f.split("/")[-3] + "/" + f.split("/")[-2]] = v
    return out or None
print("kfd evicted_ms at start:", kfd_evicted(), flush=True)
verify = os.environ.get("DF_VERIFY", "0") == "1"
b = torch.rand(N, generator=g, device="cuda", dtype=torch.float64)
worst, nbad = 0.0, 0
slowest = 0.0
t0 = time.perf_counter(); done = 0
for o in range(nobj):
    ls = LinSolverSymDense(ctx, N)
    ls.retry_after_timeout = False        # (the soak wants to SEE the time-outs)
    if os.environ.get("DF_RETRY_COPY", "0") != "1":
        ls.set_retry_copy(False)          # ... as errors; DF_RETRY_COPY=1: the shipped configuration, time-outs counted through ls.timeouts()
    for rep in range(reps):
        ls.set_sys_matrix(M); ctx.sync(); tc = time.perf_counter()
        try:
            ls.matrix_changed()
        except Exception:
            print("factorisation %d failed after %.3f s in the call" % (done, time.perf_counter() - tc), flush=True)
            print("kfd evicted_ms at the failure:", kfd_evicted(), flush=True)
            raise
        dtc = time.perf_counter() - tc
        if dtc > 0.05:
            print("factorisation %d took %.3f s" % (done, dtc), flush=True)
        slowest = max(slowest, dtc)
        done += 1
        if os.environ.get("DF_RETRY_COPY", "0") == "1" and (rep % 50 == 49 or rep == reps - 1) and ls.timeouts() > 0:
            print("time-outs absorbed by the retry copy after %d factorisations: %d" % (done, ls.timeouts()), flush=True)
            raise SystemExit(3)
        if verify:
            x = b.clone(); ls.solve(x); ctx.sync()
            res = float((M @ x - b).abs().max() / b.abs().max())
            worst = max(worst, res)
            if not res < 1e-11:
                print("factorisation %d: residual %.3e" % (done, res), flush=True); nbad += 1
    ls.close() if hasattr(ls, "close") else None
ctx.sync()
print("kfd evicted_ms at the end:", kfd_evicted(), flush=True)
print("%d factorisations of order %d without a time-out, %.2f ms each, slowest call %.1f ms%s" % (done, N, (time.perf_counter() - t0) * 1e3 / done, slowest * 1e3, (", worst solve residual %.2e, %d above 1e-11" % (worst, nbad)) if verify else ""))
