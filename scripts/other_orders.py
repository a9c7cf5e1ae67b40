"""The headline KKT step (bench.py::mds_other_orders_bench) at a list of (n_dense, n_eq) pairs: N = n_dense + n_eq + 3."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hiop_amd.runtime import Context

class A:
    ns, solves, warmup, steps = 50000, 3, 2, 10

dims = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(4097, 4093), (5000, 4200), (2500, 2497), (6000, 6286)]
ctx = Context(0)
out = bench.mds_other_orders_bench(ctx, A, dims)
for o in out["orders"]:
    print("N %6d (n_dense %5d, m %5d): %7.2f KKT iterations/s  %7.3f ms per step" % (o["N"], o["n_dense"], o["m"], o["value"], o["ms_per_step"]))
ctx.close()
