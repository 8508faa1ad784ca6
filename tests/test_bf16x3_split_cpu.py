"""The three-way bf16 split behind the fp32-accurate bf16 matrix-core kernels (csrc/pack_kernels.h: split3_pair;
csrc/conv_mfma_b3.hip header), emulated bit for bit in numpy:
    hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)           (round to nearest even, as v_cvt_pk_bf16_f32)
    a * b  ~  am*bm + al*bh + ah*bl + am*bh + ah*bm + ah*bh             (the six cross terms of weight >= 2^-16)
Pins on the CPU what the kernels assume: the split reproduces x to 24 bits, and a dot product accumulated in fp32 from
the six terms is as close to fp64 as the plain fp32 dot product is -- also for wide dynamic range and post-ReLU zeros
(the `-m gpu` adversarial test checks the kernels themselves at layer level)."""
import numpy as np


def bf16(x):
    """fp32 -> bf16 (round to nearest even) -> fp32"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = bf16(x)
    r = (x - h).astype(np.float32)
    m = bf16(r)
    s = (r - m).astype(np.float32)
    return h, m, bf16(s)


def dot6(a, b):
    """sum_k a_k b_k from the six cross terms, every product exact (bf16 x bf16 fits fp32), accumulated in fp32 small terms first"""
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    acc = np.float32(0)
    for x, y in ((am, bm), (al, bh), (ah, bl), (am, bh), (ah, bm), (ah, bh)):
        for p in (x.astype(np.float64) * y.astype(np.float64)).astype(np.float32):      # exact products, fp32 accumulation
            acc = np.float32(acc + p)
    return float(acc)


def test_split_reproduces_24_bits():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(20000) * np.exp(rng.uniform(-20, 20, 20000))).astype(np.float32)
    h, m, l = split3(x)
    rec = h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)
    err = np.abs(rec - x.astype(np.float64)) / np.abs(x.astype(np.float64))
    assert err.max() <= 2.0 ** -24, err.max()            # measured: exact for almost every input
    assert np.all(np.abs(m) <= np.abs(h) * 2.0 ** -7) and np.all(np.abs(l) <= np.abs(h) * 2.0 ** -15)


def test_zero_and_special_inputs():
    for v in (0.0, -0.0, 1.0, -1.0, 2.0 ** -126, 3.0e38):
        h, m, l = split3(np.float32(v))
        assert float(h) + float(m) + float(l) == float(np.float32(v))


def test_six_term_dot_is_fp32_accurate():
    rng = np.random.default_rng(1)
    worst6, worst32 = 0.0, 0.0
    for trial in range(20):
        n = 196 * 9                                              # the widest layer's contraction length
        a = rng.standard_normal(n) * np.exp(rng.uniform(-6, 6, n))
        a[rng.random(n) < 0.5] = 0.0                             # post-ReLU sparsity
        b = rng.standard_normal(n) * 0.05
        a32, b32 = a.astype(np.float32), b.astype(np.float32)
        ref = float(np.dot(a32.astype(np.float64), b32.astype(np.float64)))
        scale = float(np.dot(np.abs(a32).astype(np.float64), np.abs(b32).astype(np.float64)))
        acc = np.float32(0)
        for p in (a32 * b32):                                    # plain fp32: rounded products, fp32 accumulation
            acc = np.float32(acc + p)
        worst32 = max(worst32, abs(float(acc) - ref) / scale)
        worst6 = max(worst6, abs(dot6(a32, b32) - ref) / scale)
    # both are accumulation-order noise of an fp32 sum of 1764 terms; the dropped cross terms contribute < 2^-23
    assert worst6 < 2e-6 and worst32 < 2e-6, (worst6, worst32)
    assert worst6 < 4 * worst32 + 1e-7, (worst6, worst32)


def test_dropped_terms_are_below_fp32_resolution():
    rng = np.random.default_rng(2)
    a = rng.standard_normal(4096).astype(np.float32)
    b = rng.standard_normal(4096).astype(np.float32)
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    kept = sum(x.astype(np.float64) * y.astype(np.float64) for x, y in ((am, bm), (al, bh), (ah, bl), (am, bh), (ah, bm), (ah, bh)))
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(kept - exact) / np.abs(exact)
    assert rel.max() < 2.0 ** -21, rel.max()             # am*bl + al*bm + al*bl: <= 3 * 2^-8 * 2^-16 relative
