"""The debug dump of the dataflow LDL^T after an expired wait (HIOPAMD_DF_DEBUG=1): per-role / per-workgroup state
words copied by the waiter at the moment of the time-out, the host-side evaluation of what every waiting task misses, shadow flags,
publication counters, where each workgroup ran.  The wait limit is forced to 1 us in a child process so that a dump is produced.
(Named to run last: it exercises diagnostics, not the product path.)"""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

CHILD = textwrap.dedent('''
    import sys
    import torch
    sys.path.insert(0, ".")
    from hiop_amd.runtime import Context
    from hiop_amd.kkt import LinSolverSymDense
    from hiop_amd._lib import HiopAmdError
    N = 4096
    ctx = Context(0)
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    M = torch.rand(N, N, generator=g, device="cuda", dtype=torch.float64) * 1e-3
    M = M + M.T + torch.eye(N, device="cuda", dtype=torch.float64) * 10.0
    ls = LinSolverSymDense(ctx, N)
    ls.set_retry_copy(False)             # (the time-outs are to be SEEN here)
    ls.retry_after_timeout = False
    seen = 0
    for rep in range(3):
        ls.set_sys_matrix(M); ctx.sync()
        try:
            ls.matrix_changed()
        except HiopAmdError:
            seen += 1
    print("TIMEOUTS", seen)
''')


def test_debug_dump_after_a_forced_time_out(ctx):
    env = dict(os.environ, HIOPAMD_DF_TIMEOUT_MS="0.001", HIOPAMD_DF_DEBUG="1")
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert int([l for l in r.stdout.splitlines() if l.startswith("TIMEOUTS")][0].split()[1]) >= 1
    for needle in ("a bounded wait timed out", "chain role 0:", "of 480 workgroups never started", "flag words differ from their shadow copies",
                   "published their last task on another CU"):
        assert needle in r.stderr, needle + "\n" + r.stderr[-3000:]
