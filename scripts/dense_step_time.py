"""Time the dense low-rank KKT step of bench.py alone (BASELINE configs[3] shard: n_local = 1.25e6, m = 200; configs[1]: n = 1e6,
m = 100) with the roofline kernels beside it -- the A/B harness of round 4's dense-step work (scripts/calls/r04_gpu_10.sh)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hiop_amd.runtime import Context

a = argparse.Namespace(steps=int(os.environ.get("STEPS", "20")), warmup=3, solves=3, dense_nlocal=1_250_000, dense_k=200)
ctx = Context(0)
for nl, k in ((1_250_000, 200), (1_000_000, 100)):
    a.dense_nlocal, a.dense_k = nl, k
    d = bench.dense_lowrank_bench(ctx, 1, 0, a, None, rooflines=True)
    r = d["roofline"]
    print("n_local %d m %d: %.3f ms/step | gram %.3f ms (%.3f of MFMA) | J x %.3f ms (%.0f GB/s) | J^T y %.3f ms (%.0f GB/s)" % (
        nl, k, d["ms_per_step"], r[0]["avg_launch_ms"], r[0]["frac"], r[1]["avg_launch_ms"], r[1]["achieved"], r[2]["avg_launch_ms"],
        r[2]["achieved"]), flush=True)
ctx.close()
