"""hiop_amd — MI355X (gfx950) native KKT linear-system hot path behind HiOp's LinAlg / linear-solver
plug points.  The product is the C-ABI shared library `hiop_amd/lib/libhiopamd.so` (include/hiop_amd.h);
this Python package is the thin host binding used by the tests and bench (device memory via torch)."""
from ._lib import lib, check, HiopAmdError, parse_header  # noqa: F401

__version__ = "0.1.0"
