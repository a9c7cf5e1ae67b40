"""Achieved HBM bandwidth of the HBM-bound LinAlg entry points: algorithmic bytes /
time per call, HIP events on the context's stream.  Prints a markdown table (copied into profiles/<round>_probes/README.md)."""
import ctypes as C
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from bench import time_on_ctx_stream
from hiop_amd.runtime import Context

ctx = Context(0)
L = ctx._L
rows = []


def V(n, lo=0.5, hi=2.0):
    return torch.empty(n, dtype=torch.float64, device="cuda").uniform_(lo, hi)


def add(name, ref, nbytes, fn, reps=20):
    ms = time_on_ctx_stream(ctx, fn, reps)
    rows.append((name, ref, nbytes / 1e6, ms * 1e3, nbytes / (ms * 1e-3) / 1e9))


for n in (10_000_000, 1_000_000, 100_000):
    x, y, z, sel = V(n), V(n), V(n), (V(n) > 1.0).double()
    tag = f"n={n:.0e}"
    add(f"vec_axpy {tag}", "hiopVectorPar.cpp:664", 24 * n, lambda: ctx.call("hiopamd_vec_axpy", n, y, 0.5, x))
    add(f"vec_axdzpy_w_pattern {tag}", ":760", 40 * n, lambda: ctx.call("hiopamd_vec_axdzpy_w_pattern", n, y, 0.5, x, z, sel))
    add(f"vec_component_mult {tag}", ":564", 24 * n, lambda: ctx.call("hiopamd_vec_component_mult", n, y, x))
    add(f"vec_copy {tag}", ":153", 16 * n, lambda: ctx.call("hiopamd_vec_copy", n, y, x))
    out = C.c_double(0)
    add(f"vec_dot {tag} (returns to host)", ":480", 16 * n, lambda: ctx.call("hiopamd_vec_dot", n, x, y, C.byref(out)))
    add(f"vec_log_barrier {tag} (returns to host)", ":863", 16 * n, lambda: ctx.call("hiopamd_vec_log_barrier", n, x, sel, C.byref(out)))
    add(f"vec_fraction_to_the_bdry {tag} (returns to host)", ":1009", 16 * n,
        lambda: ctx.call("hiopamd_vec_fraction_to_the_bdry", n, x, y, 0.99, C.byref(out)))

for k, n in ((200, 1_250_000), (100, 1_000_000), (8, 10_000_000)):
    J = torch.empty((k, n), dtype=torch.float64, device="cuda").uniform_(-1, 1)
    xv, yv = V(n), V(k)
    tag = f"{k}x{n:.2e}"
    add(f"mat_times_vec {tag}", "hiopMatrixDenseRowMajor.cpp:458", 8 * (k * n + n + k), lambda: ctx.call("hiopamd_mat_times_vec", k, n, J, n, 0.0, yv, 1.0, xv), 10)
    add(f"mat_trans_times_vec {tag}", ":510", 8 * (k * n + 2 * n + k), lambda: ctx.call("hiopamd_mat_trans_times_vec", k, n, J, n, 1.0, xv, 1.0, yv), 10)
    del J

# row-sorted COO SpMV (MDS sparse Jacobian shapes: few entries per row, wide)
r = np.random.Generator(np.random.PCG64(1))
for m, n, per_row in ((4096, 100_000, 3), (4096, 1_000_000, 30), (100_000, 100_000, 5)):
    i = np.repeat(np.arange(m, dtype=np.int32), per_row)
    j = np.sort(r.integers(0, n, (m, per_row)), axis=1).astype(np.int32).ravel()
    nnz = i.size
    id_, jd = torch.as_tensor(i).cuda(), torch.as_tensor(j).cuda()
    vd, xv, yv = V(nnz), V(n), V(m)
    tag = f"{m}x{n:.0e}, nnz={nnz:.1e}"
    add(f"sp_times_vec {tag}", "hiopMatrixSparseTriplet.cpp:73", 16 * nnz + 8 * nnz + 8 * m, lambda: ctx.call("hiopamd_sp_times_vec", m, n, nnz, id_, jd, vd, 0.0, yv, 1.0, xv), 20)
    add(f"sp_trans_times_vec {tag}", ":110", 16 * nnz + 16 * nnz + 8 * n, lambda: ctx.call("hiopamd_sp_trans_times_vec", m, n, nnz, id_, jd, vd, 0.0, xv, 1.0, yv), 20)

print("| entry point | reference | algorithmic MB | us per call | GB/s | of 8 TB/s |")
print("|---|---|---:|---:|---:|---:|")
for name, ref, mb, us, gbs in rows:
    print(f"| {name} | {ref} | {mb:.1f} | {us:.1f} | {gbs:.0f} | {gbs / 8000:.2f} |")
json.dump([dict(name=a, ref=b, MB=c, us=d, GBs=e) for a, b, c, d, e in rows], open("gpurun_out/bw_table.json", "w"), indent=1)
