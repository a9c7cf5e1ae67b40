"""The factorisation family and the C interface on the POISON build of the library (hiop_amd/build.py, HIOPAMD_BUILD_VARIANT=poison:
every device allocation the library makes is filled with 0xFF bytes — NaN as doubles, -1 as ints — before anybody writes it).

Why this is part of the gate: round 5's driver run failed at its first GPU test with a search direction full of NaN at N = 503 while
the same binary passed 418 tests on other boxes.  The block inversion behind a factorisation read the padding columns of the ragged
last compact diagonal block, which no kernel had ever written: 0 x (whatever the fresh HBM held) — 0 on most boxes, NaN on that one.
A thousand fresh-process solves on the round-5 library did not reproduce it; the poison build reproduces it in every run
(profiles/r06_probes/call01_*).  A kernel that reads memory nobody wrote now fails HERE, on every box, not on one box in twenty.

The nested runs are ordinary pytest runs of the named files in a child process whose environment selects the poison library."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _poison_env():
    env = dict(os.environ)
    env["HIOPAMD_BUILD_VARIANT"] = "poison"
    return env


@pytest.fixture(scope="module")
def poison_lib():
    r = subprocess.run([sys.executable, "-m", "hiop_amd.build"], cwd=ROOT, env=_poison_env(), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = ROOT / "hiop_amd" / "lib_poison" / "libhiopamd.so"
    assert lib.exists()
    return lib


@pytest.mark.parametrize("files", [
    ["tests/test_c_interface.py"],
    ["tests/test_gpu_ldlt_kkt.py", "tests/test_ldlt_exact_closed_form.py"],
    ["tests/test_gpu_ldlt_bk.py", "tests/test_gpu_sparse_ldl.py"],
    ["tests/test_gpu_kkt_xycyd.py", "tests/test_gpu_lowrank.py"],
], ids=["c_interface", "ldlt_kkt", "ldlt_bk+sparse_ldl", "kkt_xycyd+lowrank"])
def test_suite_passes_with_every_library_allocation_poisoned(poison_lib, files):
    r = subprocess.run([sys.executable, "-m", "pytest", *files, "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"], cwd=ROOT, env=_poison_env(),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1]
    # the child really loaded the poison build (a silent fall-back to the shipped library would prove nothing)
    chk = subprocess.run([sys.executable, "-c", "from hiop_amd._lib import LIBPATH; print(LIBPATH)"], cwd=ROOT, env=_poison_env(),
                         capture_output=True, text=True)
    assert chk.stdout.strip() == str(poison_lib)
