"""The IDCT kernel multiplies by the reference's double constants in single precision.

dct.c:296-341 forms products such as ``1.414213562 * x`` with x a float: the product is
rounded to double and then to float.  The kernel computes fma(hi, x, lo * x) with
hi = (float) C and lo = (float) (C - hi) instead (two packed instructions for two values).
That is only allowed if it yields the same float for every x, which is a property of each
constant: checked here over all 2^23 mantissas of x (signs and exponents do not change the
rounding pattern away from the subnormal range, which 8-bit-sample arithmetic never
reaches)."""
import numpy as np
import pytest


@pytest.mark.parametrize("c", [0.707106781, 1.414213562, 2.0 * 0.707106781, 0.25 * 0.707106781])
def test_split_constant_product_equals_the_double_product(c):
    c = np.float64(c)
    hi = np.float32(c)
    lo = np.float32(c - np.float64(hi))
    for exponent in (0, 9):
        x = (np.arange(1 << 23, dtype=np.uint32) + np.uint32((127 + exponent) << 23)).view(np.float32)
        want = (c * x.astype(np.float64)).astype(np.float32)
        low = (lo * x).astype(np.float32)                   # float32 product, rounded once
        # fma: hi * x is exact in 48 bits and low lies at most 26 bits below its top bit, so the
        # 64-bit significand of the extended type holds the sum exactly; one rounding to float
        wide = hi.astype(np.longdouble) * x.astype(np.longdouble) + low.astype(np.longdouble)
        assert np.finfo(np.longdouble).nmant >= 63
        got = wide.astype(np.float32)
        assert np.array_equal(got, want)
        assert np.array_equal((-wide).astype(np.float32), (c * (-x).astype(np.float64)).astype(np.float32))
