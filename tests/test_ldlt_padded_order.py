"""The order a dense solver object works at (csrc/ldlt.hip::ldlt_padded_order, round 6): the kernels' fast forms take even orders
(16-byte tile form), multiples of 256 (the whole factorisation as one dataflow) and multiples of 512 (the solve's 512-row blocks); every
other order n >= 1024 is factored and solved as diag(M, I) of a padded order chosen by a cost model.  Host logic: no device needed."""
import os
import subprocess
import sys

import pytest

from hiop_amd._lib import lib


def order(n):
    return lib().hiopamd_ldlt_padded_order(n)


def test_invariants_over_every_order():
    for n in list(range(0, 3000)) + list(range(3000, 21000, 7)) + [8191, 8192, 8193, 16383, 16384, 16385, 20479, 20480]:
        c = order(n)
        if n < 1024:
            assert c == n                       # small orders run the stepwise kernels as they are
            continue
        assert n <= c <= (n + 511) // 512 * 512, (n, c)
        assert c % 2 == 0, (n, c)               # never the 8-byte tile form
        assert c in (n + (n & 1), (n + 255) // 256 * 256, (n + 511) // 512 * 512), (n, c)
        if n % 512 == 0:
            assert c == n                       # the best case stays what it is


def test_the_choices_the_measurements_were_taken_at():
    # profiles/r06_probes/call24_padded_orders_cost_model_vs_parity_only.txt: the model's choice was the faster (or equal) one at every order
    want = {1025: 1280, 2000: 2048, 2047: 2048, 3000: 3072, 4000: 4096, 4097: 4608, 5000: 5120, 6000: 6144, 7000: 7168, 8000: 8192,
            8003: 8192, 8191: 8192, 8192: 8192, 8193: 8194, 8500: 8704, 9000: 9216, 10000: 10240, 12289: 12290, 16383: 16384, 6503: 6656}
    got = {n: order(n) for n in want}
    assert got == want


def test_environment_switch():
    code = "from hiop_amd._lib import lib; L = lib(); print(L.hiopamd_ldlt_padded_order(8003), L.hiopamd_ldlt_padded_order(8000))"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode, want in (("0", "8003 8000"), ("1", "8004 8000"), ("2", "8192 8192")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, env=dict(os.environ, HIOPAMD_LDLT_PAD=mode), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.split()[-2:] == want.split(), (mode, r.stdout)
