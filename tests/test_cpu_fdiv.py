"""The float-reciprocal index decodes of the kernels' prologues (csrc/common.h: fdiv / fdivmod_px), restated in float32 NumPy arithmetic and
checked against integer division over their documented domains - including a reciprocal that is one ulp off either way (v_rcp_f32's bound).
fdiv:        floor(n / d) == int(float32(2n + 1) * r),  r = 0.5 * rcp(d),          0 <= n < 2^20
fdivmod_px:  q = int(float32(n) * rcp(d)) corrected by one step either way,         0 <= n < 2^23"""
import numpy as np


def _cases(nmax, rng):
    ds = np.unique(np.concatenate([np.arange(1, 700), rng.integers(1, nmax, 300), 2 ** np.arange(0, 20), 2 ** np.arange(1, 20) - 1, 2 ** np.arange(1, 20) + 1]))
    for d in ds:
        d = int(d)
        k = np.arange(0, nmax // d + 1, max(1, (nmax // d) // 4000), dtype=np.int64) * d          # multiples of d and their neighbours
        n = np.unique(np.concatenate([k - 1, k, k + 1, rng.integers(0, nmax, 2000), [0, nmax - 1]]))
        yield d, n[(n >= 0) & (n < nmax)]


def _rcps(d):
    r0 = np.float32(1.0) / np.float32(d)
    return [r0, np.nextafter(r0, np.float32(0)), np.nextafter(r0, np.float32(np.inf))]


def test_fdiv_is_exact_below_2_pow_20():
    rng = np.random.default_rng(0)
    for d, n in _cases(1 << 20, rng):
        for rc in _rcps(d):
            r = np.float32(0.5) * rc
            q = (np.float32(1) * (2 * n + 1).astype(np.float32) * r).astype(np.int64)        # float32 RNE multiply, truncation
            assert np.array_equal(q, n // d), (d, float(rc))


def test_fdivmod_px_is_exact_below_2_pow_23():
    rng = np.random.default_rng(1)
    for d, n in _cases(1 << 23, rng):
        for rc in _rcps(d):
            q = (n.astype(np.float32) * rc).astype(np.int64)
            rem = n - q * d
            q = np.where(rem < 0, q - 1, q); rem = np.where(rem < 0, rem + d, rem)
            q = np.where(rem >= d, q + 1, q); rem = np.where(rem >= d, rem - d, rem)
            assert np.array_equal(q, n // d) and np.array_equal(rem, n % d), (d, float(rc))
