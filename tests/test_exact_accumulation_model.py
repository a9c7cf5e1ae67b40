"""CPU model of the order-independent accumulation behind the transposed / symmetric sparse products
(hiop_amd/csrc/sparse_kernels.hip: exact_add / exact_value, round 4): every contribution is truncated to a multiple of the quantum
2^(E - 93) (E = exponent of the largest finite contribution), added as a signed integer spread over three 32-bit limbs that live in
64-bit words (so that plain 64-bit integer atomics can carry), and the total is rounded to a double once.  Restated here with Python
integers, limb for limb, to check what the design promises:
  * the words after any permutation of the contributions are identical (integer addition modulo 2^64 per word commutes);
  * the value equals the exactly rounded sum whenever no term is smaller than 2^-93 of the largest one -- in particular it is at
    least as accurate as any chain of rounded additions;
  * 2^32 contributions of the largest magnitude cannot overflow the 96-bit total;
  * cancellation to zero gives +0.0, signs and subnormals are handled.
The bitwise run-to-run reproducibility of the HIP kernels themselves is tested on the GPU
(tests/test_gpu_dense_sparse.py::test_transposed_and_symmetric_products_are_bitwise_reproducible)."""
import math
import struct
from fractions import Fraction

import numpy as np
import pytest

M64 = (1 << 64) - 1


def bits_of(p):
    return struct.unpack("<Q", struct.pack("<d", p))[0]


def ilogb(bits):
    e = (bits >> 52) & 0x7FF
    return e - 1023 if e else -1022


def exact_add(acc, p, eq):
    """acc: list of four words (mod 2^64).  Mirrors exact_add() of sparse_kernels.hip."""
    bits = bits_of(p)
    eb = (bits >> 52) & 0x7FF
    m = bits & 0x000FFFFFFFFFFFFF
    if eb == 0x7FF:
        acc[3] |= 4 if m else (2 if bits >> 63 else 1)
        return
    if eb:
        m |= 0x0010000000000000
    if m == 0:
        return
    sh = (eb - 1023 if eb else -1022) - 52 - eq
    if sh >= 0:
        if sh >= 64:
            lo, hi = 0, (m << (sh - 64)) & M64
        else:
            lo, hi = (m << sh) & M64, ((m >> (64 - sh)) if sh else 0)
    else:
        if sh <= -53:
            return
        lo, hi = m >> (-sh), 0
    if bits >> 63:
        lo = (~lo + 1) & M64
        hi = (~hi + (1 if lo == 0 else 0)) & M64
    l0, l1 = lo & 0xFFFFFFFF, lo >> 32
    h32 = hi & 0xFFFFFFFF
    l2 = (h32 - (1 << 32) if h32 & 0x80000000 else h32) & M64     # sign-extended low 32 bits of hi, as a 64-bit word
    acc[0] = (acc[0] + l0) & M64
    acc[1] = (acc[1] + l1) & M64
    acc[2] = (acc[2] + l2) & M64


def exact_value(acc, eq):
    t = (acc[0] + (acc[1] << 32) + (acc[2] << 64)) & ((1 << 128) - 1)
    st = t - (1 << 128) if t >> 127 else t
    if st == 0:
        return 0.0
    mag = abs(st)
    nbits = mag.bit_length()
    if nbits > 64:                     # the leading 64 bits, the rest dropped (deterministically), then one correct rounding
        top, shift = mag >> (nbits - 64), nbits - 64
    else:
        top, shift = mag << (64 - nbits), -(64 - nbits)
    v = math.ldexp(float(top), shift + eq)        # float(int) is correctly rounded
    return -v if st < 0 else v


def accumulate(terms, order=None):
    finite = [abs(t) for t in terms if math.isfinite(t)]
    mb = max((bits_of(a) for a in finite), default=0)
    eq = ilogb(mb) - 93
    acc = [0, 0, 0, 0]
    for k in (order if order is not None else range(len(terms))):
        exact_add(acc, terms[k], eq)
    return acc, eq


def rounded_exact_sum(terms):
    f = sum((Fraction(t) for t in terms), Fraction(0))
    return float(f)                                   # Fraction -> float rounds correctly


@pytest.mark.parametrize("seed", range(6))
def test_any_order_gives_the_same_words_and_the_correctly_rounded_sum(seed):
    r = np.random.default_rng(seed)
    n = 2000
    terms = list((r.uniform(-1, 1, n) * 10.0 ** r.integers(-6, 7, n)).astype(float))    # 12 decades: everything above 2^-93 of the maximum
    acc0, eq = accumulate(terms)
    for _ in range(4):
        acc, _ = accumulate(terms, order=list(r.permutation(n)))
        assert acc == acc0
    v = exact_value(acc0, eq)
    assert v == rounded_exact_sum(terms)
    # a chain of rounded additions is not reproducible under permutation, and not more accurate
    chain = [float(np.sum(np.array(terms)[r.permutation(n)], dtype=np.float64)) for _ in range(3)]
    assert any(abs(c - rounded_exact_sum(terms)) >= abs(v - rounded_exact_sum(terms)) for c in chain)


def test_cancellation_signs_and_tiny_terms():
    terms = [1.0, 1e-30, -1.0]                       # the tiny term is below 2^-93 of the maximum: dropped, by design
    acc, eq = accumulate(terms)
    assert exact_value(acc, eq) == 0.0 and math.copysign(1.0, exact_value(acc, eq)) == 1.0
    terms = [3.5, -1.25, 2.0 ** -40, -(2.0 ** -41)]
    acc, eq = accumulate(terms)
    assert exact_value(acc, eq) == rounded_exact_sum(terms)
    sub = 5e-324
    acc, eq = accumulate([sub, sub, sub])
    assert exact_value(acc, eq) == 3 * sub           # subnormals: exponent -1022, no hidden bit


def test_two_to_the_32_largest_terms_do_not_overflow():
    """|term| < 2^94 quanta; 2^32 of them stay below 2^126 < 2^127: the signed 128-bit view of the three words is exact.  Checked by
    adding one term's limbs 2^32 times arithmetically (the limbs of equal terms are equal)."""
    p = 1.9999999999999998                            # mantissa all ones: the largest magnitude for its exponent
    acc, eq = accumulate([p])
    cnt = 1 << 32
    big = [(acc[0] * cnt) & M64, (acc[1] * cnt) & M64, (acc[2] * cnt) & M64, 0]
    assert exact_value(big, eq) == p * cnt            # (a power-of-two multiple: exactly representable)
    accn, eqn = accumulate([-p])
    bign = [(accn[0] * cnt) & M64, (accn[1] * cnt) & M64, (accn[2] * cnt) & M64, 0]
    assert exact_value(bign, eqn) == -p * cnt


def test_non_finite_contributions_are_flagged_per_output():
    acc, eq = accumulate([1.0, float("inf"), -2.0])
    assert acc[3] == 1
    acc, eq = accumulate([float("-inf"), float("nan")])
    assert acc[3] == 2 | 4
