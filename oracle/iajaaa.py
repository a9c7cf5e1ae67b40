"""Reader for HiOp's `.iajaaa` linear-system dumps — TEST INFRASTRUCTURE.
Format: src/LinAlg/csr_iajaaa.md:9-29 (writer: src/Utils/hiopCSR_IO.hpp:89-152)."""
from __future__ import annotations

import numpy as np


def read_iajaaa(path):
    toks = open(path).read().split()
    pos = 0

    def take(n, conv):
        nonlocal pos
        out = [conv(t) for t in toks[pos:pos + n]]
        pos += n
        return out

    nrows, = take(1, int)
    nx, neq, nineq = take(3, int)
    nnz, = take(1, int)
    ia = np.array(take(nrows + 1, int)) - 1
    ja = np.array(take(nnz, int)) - 1
    aa = np.array(take(nnz, float))
    M = np.zeros((nrows, nrows))
    for r in range(nrows):
        M[r, ja[ia[r]:ia[r + 1]]] = aa[ia[r]:ia[r + 1]]
    pairs = []
    while pos + 2 * nrows <= len(toks):
        rhs = np.array(take(nrows, float))
        sol = np.array(take(nrows, float))
        pairs.append((rhs, sol))
    return dict(n=nrows, nx=nx, neq=neq, nineq=nineq, M_upper=M, pairs=pairs)
