#!/usr/bin/env python3
"""Measurement yardstick ONLY (never linked into the product): what does the vendor's DGEMM (rocBLAS/hipBLASLt behind
torch.mm, fp64) sustain on this MI355X for the shapes of the LDL^T trailing update?

  C(8192 x 8192) -= A^T(8192 x K) B(K x 8192)   for K = 256 (one super-panel update) and K = 8192 (large-K asymptote)

Prints TFLOP/s per shape; run under `rocprofv3 --kernel-trace --stats` to get the kernel names (Tensile names carry the
macro-tile and the MFMA instruction, e.g. MT128x128x16_MI16x16x4x1 = v_mfma_f64_16x16x4_f64).  The gfx950 Tensile
DGEMM code objects of the image contain only v_mfma_f64_16x16x4_f64 (profiles/r02_probes/README.md, llvm-objdump)."""
import json
import sys

import torch


def bench(m, n, k, reps=10):
    a = torch.randn(k, m, dtype=torch.float64, device="cuda")   # K-major, like the row panels of the factorisation
    b = torch.randn(k, n, dtype=torch.float64, device="cuda")
    c = torch.randn(m, n, dtype=torch.float64, device="cuda")
    for _ in range(2):
        c.addmm_(a.t(), b, beta=1.0, alpha=-1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        c.addmm_(a.t(), b, beta=1.0, alpha=-1.0)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return dict(m=m, n=n, k=k, ms=ms, tflops=2.0 * m * n * k / (ms * 1e-3) / 1e12)


if __name__ == "__main__":
    out = [bench(8192, 8192, 256), bench(8192, 8192, 8192, reps=3), bench(4096, 4096, 256), bench(8192, 8192, 64)]
    for o in out:
        print(json.dumps(o))
    json.dump(out, open(sys.argv[1], "w"), indent=1) if len(sys.argv) > 1 else None
