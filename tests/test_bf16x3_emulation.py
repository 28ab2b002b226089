"""The arithmetic of csrc/bf16x3.h restated in NumPy (CPU, no GPU needed): the three-way bf16 split is exact, every
bf16 x bf16 product is exact in fp32, and the six-product sum reproduces an fp32 dot product to fp32 rounding - the claims
the GRU cell / dense-layer kernels (csrc/gru_x3.hip, csrc/gemm_x3.hip) rest on.  The GPU tests check the kernels against
float64 (`test_gemm_bf16x3_vs_float64`, `test_bf16x3_split_is_exact`); this file pins the error analysis itself."""
import numpy as np


def bf16_round(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32: what v_cvt_pk_bf16_f32 does."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(a):
    a = np.asarray(a, dtype=np.float32)
    a1 = bf16_round(a)
    r1 = (a - a1).astype(np.float32)          # exact in fp32 (Sterbenz-like: a1 is a's own leading bits)
    a2 = bf16_round(r1)
    r2 = (r1 - a2).astype(np.float32)
    a3 = bf16_round(r2)
    return a1, a2, a3


def test_three_way_split_is_exact_over_the_fp32_range():
    rng = np.random.default_rng(0)
    mag = np.exp2(rng.integers(-60, 60, 200000)).astype(np.float32)
    a = (rng.standard_normal(200000).astype(np.float32) * mag)
    a = np.concatenate([a, np.float32([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.17549435e-38])])
    a1, a2, a3 = split3(a)
    total = a1.astype(np.float64) + a2.astype(np.float64) + a3.astype(np.float64)
    assert np.array_equal(total, a.astype(np.float64))            # a1 + a2 + a3 == a, bit for bit
    nz = a != 0
    assert np.all(np.abs(a2[nz]) <= np.abs(a[nz]) * 2.0 ** -8) and np.all(np.abs(a3[nz]) <= np.abs(a[nz]) * 2.0 ** -16)


def test_bf16_products_are_exact_in_fp32():
    rng = np.random.default_rng(1)
    x = bf16_round(rng.standard_normal(100000).astype(np.float32))
    y = bf16_round(rng.standard_normal(100000).astype(np.float32))
    p32 = (x * y).astype(np.float32)                                # 8 x 8 significand bits fit fp32's 24
    assert np.array_equal(p32.astype(np.float64), x.astype(np.float64) * y.astype(np.float64))


def test_six_products_reproduce_an_fp32_dot_product():
    """sum_k a_k b_k via the six kept products (fp32 accumulation, smallest first) against float64: the error relative to
    sum_k |a_k b_k| is at the level of an fp32 accumulation of the same length, and the three dropped products are below
    2^-23 |a b| each."""
    rng = np.random.default_rng(2)
    K, rows = 576, 512                                              # the GRU cell's contraction length
    a = rng.standard_normal((rows, K)).astype(np.float32)
    b = (rng.standard_normal((rows, K)) * 0.06).astype(np.float32)
    sa, sb = split3(a), split3(b)
    kept = [(0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)]          # (a1 b3) (a3 b1) (a2 b2) (a1 b2) (a2 b1) (a1 b1)
    acc = np.zeros(rows, dtype=np.float32)
    for i, j in kept:                                               # one MFMA per product: fp32 accumulate over K
        for k0 in range(0, K, 32):
            acc = (acc + (sa[i][:, k0:k0 + 32] * sb[j][:, k0:k0 + 32]).astype(np.float32).sum(1, dtype=np.float32)).astype(np.float32)
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    scale = np.abs(a.astype(np.float64) * b.astype(np.float64)).sum(1)
    err = np.abs(acc.astype(np.float64) - ref) / scale
    plain = np.abs((a * b).astype(np.float32).sum(1, dtype=np.float32).astype(np.float64) - ref) / scale   # fp32 products + fp32 sum
    assert err.max() < 4e-7 and err.mean() < 5e-8, (err.max(), err.mean())
    assert err.mean() < 3.0 * plain.mean() + 1e-9                    # no worse than an fp32 pipeline in any meaningful sense
    dropped = sum(np.abs(sa[i].astype(np.float64) * sb[j].astype(np.float64)) for i, j in [(1, 2), (2, 1), (2, 2)])
    assert np.all(dropped <= 2.0 ** -23 * np.abs(a.astype(np.float64) * b.astype(np.float64)) + 1e-300)
