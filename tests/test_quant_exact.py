"""The HIP quantiser evaluates sign(t)*floor((|t|+4q)/8q) as trunc(t*r +- b) in fp32 (mij_kernels.hip quant1).
Brute force: exact for every divisor 8q (q = 1..255) and every DCT output the 8-bit path can produce."""
import numpy as np


def test_float_reciprocal_quantiser_is_exact():
    t = np.arange(-40000, 40001, dtype=np.int64)
    tf = t.astype(np.float32)
    for q in range(1, 256):
        d = 8 * q
        ref = np.sign(t) * ((np.abs(t) + d // 2) // d)
        r = np.float32(1.0) / np.float32(d)
        b = (np.float32(4 * q) + np.float32(0.5)) * r
        bs = np.copysign(b, tf).astype(np.float32)
        unfused = np.trunc(tf * r + bs).astype(np.int64)
        fused = np.trunc((tf.astype(np.float64) * np.float64(r) + bs.astype(np.float64)).astype(np.float32)).astype(np.int64)
        assert np.array_equal(unfused, ref), q
        assert np.array_equal(fused, ref), q
