"""The arithmetic claim behind csrc/bf16x3.h, checked on the CPU with numpy (no GPU needed):

* an f32 value splits EXACTLY into three bf16 terms (round to nearest even), x = h + m + l;
* the six products the kernels keep (all but m*l, l*m, l*l) reproduce a*b to 2^-24 |a b|, i.e.
  to f32's own rounding, for every operand magnitude f32 can hold without under/overflow of the
  terms.

This is a restatement of the device code's arithmetic, not of its scheduling; the GPU parity tests
(tests/test_gpu_*.py) cover the kernels themselves."""
import numpy as np


def bf16_rne(x: np.ndarray) -> np.ndarray:
    """f32 -> bf16 -> f32, round to nearest even (what v_cvt_pk_bf16_f32 does for finite values)."""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (r << 16).astype(np.uint32).view(np.float32)


def split3(x: np.ndarray):
    x = x.astype(np.float32)
    h = bf16_rne(x)
    r1 = (x - h).astype(np.float32)
    m = bf16_rne(r1)
    r2 = (r1 - m).astype(np.float32)
    l = bf16_rne(r2)
    return h, m, l


def samples(n=200000, seed=0, max_exp=60):
    rng = np.random.default_rng(seed)
    mant = rng.uniform(1.0, 2.0, n).astype(np.float32)
    expo = rng.integers(-max_exp, max_exp, n)
    sign = rng.choice([-1.0, 1.0], n).astype(np.float32)
    x = (sign * np.ldexp(mant, expo)).astype(np.float32)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.0 / 256, 3.0e38,
                     1.0e-30, 0.1, 1.0 / 3, 16777215.0, 1.99999988], dtype=np.float32)
    return np.concatenate([x, edge])


def test_three_bf16_terms_hold_an_f32_exactly():
    x = samples()
    h, m, l = split3(x)
    # float64 holds the sum of three f32 values of these magnitudes exactly
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64),
                          x.astype(np.float64))
    # each term really is a bf16 (low 16 bits of the f32 pattern are zero)
    for t in (h, m, l):
        assert not np.any(t.view(np.uint32) & 0xFFFF)
    # and the terms shrink by 2^-8 each, which is what bounds the dropped products
    nz = x != 0
    assert np.all(np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8)
    assert np.all(np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_six_products_reproduce_the_f32_product():
    # exponents within +-40: the smallest kept product (2^-16 |a b|) then stays a normal f32
    a, b = samples(seed=1, max_exp=40), samples(seed=2, max_exp=40)[::-1].copy()
    keep = (np.abs(a) < 1e30) & (np.abs(b) < 1e30) & ((a == 0) | (np.abs(a) > 1e-20)) & ((b == 0) | (np.abs(b) > 1e-20))
    a, b = a[keep], b[keep]
    ah, am, al = (t.astype(np.float64) for t in split3(a))
    bh, bm, bl = (t.astype(np.float64) for t in split3(b))
    kept = al * bh + ah * bl + am * bm + am * bh + ah * bm + ah * bh   # exact in float64
    exact = a.astype(np.float64) * b.astype(np.float64)
    err = np.abs(kept - exact)
    assert np.all(err <= np.abs(exact) * 2.0 ** -24)
    # every kept product of two bf16 values is exactly representable in f32 (what the MFMA multiplies)
    for p in (al * bh, ah * bl, am * bm, am * bh, ah * bm, ah * bh):
        assert np.array_equal(p.astype(np.float32).astype(np.float64), p)


def test_dot_product_matches_f32_accumulation_quality():
    """K = 256 dot products: the six-product form summed in f32 is as close to float64 as a plain
    f32 dot product (the device probe tools/probes/bf16x3_probe.hip measures the same on the MFMA)."""
    rng = np.random.default_rng(5)
    a = rng.uniform(-1, 1, (512, 256)).astype(np.float32)
    b = rng.uniform(-1, 1, (512, 256)).astype(np.float32)
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(axis=1)
    mag = np.abs(a.astype(np.float64) * b.astype(np.float64)).sum(axis=1)
    plain = np.zeros(512, np.float32)
    x3 = np.zeros(512, np.float32)
    sa, sb = split3(a), split3(b)
    for k in range(256):
        plain = (plain + a[:, k] * b[:, k]).astype(np.float32)
        for i, j in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):
            x3 = (x3 + sa[i][:, k] * sb[j][:, k]).astype(np.float32)
    e_plain = np.sqrt(np.mean(((plain - ref) / mag) ** 2))
    e_x3 = np.sqrt(np.mean(((x3 - ref) / mag) ** 2))
    assert e_x3 <= 2.0 * e_plain and e_x3 < 1e-7
