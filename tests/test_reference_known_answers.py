"""Known answers of the reference's own LinAlg unit tests (tests/golden/reference_unit_tests.json):
 * CPU: they pin the ORACLE (every case must reproduce the reference's expected value);
 * GPU: the HIP kernels must reproduce the same values through the C ABI."""
import pytest

from ref_cases import check, load_cases, run_gpu, run_oracle

CASES = load_cases()
IDS = [f"{i}-{c['op']}" for i, c in enumerate(CASES)]


def test_fixture_is_reproducible():
    """The committed JSON is exactly what the committed generator produces."""
    import json, subprocess, sys, tempfile, shutil, os
    from pathlib import Path
    gold = Path(__file__).parent / "golden"
    before = (gold / "reference_unit_tests.json").read_text()
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(gold / "make_reference_unit_tests.py", td)
        subprocess.check_call([sys.executable, os.path.join(td, "make_reference_unit_tests.py")], stdout=subprocess.DEVNULL)
        assert Path(td, "reference_unit_tests.json").read_text() == before


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_reproduces_reference_known_answer(case):
    check(case, run_oracle(case))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_hip_reproduces_reference_known_answer(ctx, case):
    check(case, run_gpu(ctx, case))
