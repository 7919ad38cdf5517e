"""CPU: the arithmetic claims behind the conv stack's split-emulation modes (csrc/scnet.hip SPLIT 4 / 5), checked in numpy.

  * three bfloat16 pieces (round-to-nearest-even, as v_cvt_pk_bf16_f32 and the host packer round) hold a float32 EXACTLY: a == a1 + a2 + a3;
  * every partial product ai * bj is exact in float32 (8 x 8 significand bits);
  * bf16x9: the nine partial products sum to a * b exactly (in exact arithmetic: float64 holds all of it);
  * bf16x6: what the six kept terms miss is at most 2^-23 |a b| (|a2| <= 2^-8 |a|, |a3| <= 2^-16 |a|: two terms of <= 2^-24 and one of 2^-32), in practice
    below 2^-24 with an rms a quarter of the rms rounding error of a correctly rounded float32 multiply (bound 2^-24 |a b|);
  * the two-piece modes for comparison: bf16x3 misses ~2^-16, f16x3 (22-bit operands) ~2^-21.
Reference: the fp32 convolutions of model/mymodel.py:15-39 that these modes emulate."""
import numpy as np


def bf16(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    a1 = bf16(x)
    r1 = x - a1                      # exact in float32
    a2 = bf16(r1)
    r2 = r1 - a2                     # exact
    a3 = bf16(r2)
    return a1, a2, a3, r2 - a3


def _samples(n, seed):
    rs = np.random.RandomState(seed)
    x = (rs.standard_normal(n) * np.exp(rs.uniform(-20, 20, n))).astype(np.float32)
    edge = np.array([1.0, -1.0, 1 + 2.0 ** -23, 1 - 2.0 ** -24, 3.1415927, 0.1, 65504.0, 1e-30, -7.0e20, 2.0 ** -100, 0.0], dtype=np.float32)
    return np.concatenate((x, edge))


def test_three_bf16_pieces_hold_a_float32_exactly():
    x = _samples(200000, 1)
    a1, a2, a3, rest = split3(x)
    assert not rest.any()                                                   # nothing left after the third piece
    assert np.array_equal((a1.astype(np.float64) + a2 + a3).astype(np.float32), x)
    assert np.array_equal(a1.astype(np.float64) + a2.astype(np.float64) + a3.astype(np.float64), x.astype(np.float64))
    for p in (a1, a2, a3):                                                  # each piece IS a bfloat16 (low 16 bits clear)
        assert not (p.view(np.uint32) & 0xFFFF).any()
    # magnitudes: |a2| <= 2^-8 |a1| (half an ulp of 8 bits), |a3| <= 2^-16 |a1|
    nz = a1 != 0
    assert (np.abs(a2[nz]) <= np.abs(a1[nz]) * 2.0 ** -8).all() and (np.abs(a3[nz]) <= np.abs(a1[nz]) * 2.0 ** -16).all()


def test_partial_products_are_exact_and_nine_of_them_are_the_product():
    a, b = _samples(100000, 2), _samples(100000, 3)[::-1].copy()
    A, B = split3(a)[:3], split3(b)[:3]
    exact = a.astype(np.float64) * b.astype(np.float64)                     # 48 significand bits: exact in float64
    tot = np.zeros_like(exact)
    for ai in A:
        for bj in B:
            p32 = ai * bj                                                   # float32 product of two bfloat16
            p64 = ai.astype(np.float64) * bj.astype(np.float64)
            ok = np.isfinite(p32) & ((np.abs(p64) >= 2.0 ** -126) | (p64 == 0))    # (leave float32 overflow / the subnormal range of the extreme samples aside)
            assert np.array_equal(p32[ok].astype(np.float64), p64[ok])      # 16 significand bits: exact in float32
            tot += p64
    assert np.array_equal(tot, exact)                                       # bf16x9: the nine terms ARE a * b


def test_bf16x6_misses_less_than_a_float32_multiply_rounds():
    a, b = _samples(100000, 4), _samples(100000, 5)
    A, B = split3(a)[:3], split3(b)[:3]
    exact = a.astype(np.float64) * b.astype(np.float64)
    kept = sum(A[i].astype(np.float64) * B[j].astype(np.float64) for i, j in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)))
    nz = exact != 0
    rel6 = np.abs(kept[nz] - exact[nz]) / np.abs(exact[nz])
    assert rel6.max() <= 2.0 ** -23                                         # bf16x6: dropped a2 b3 + a3 b2 + a3 b3 -- the analytic worst case
    assert rel6.max() <= 2.0 ** -24                                         # ... and what random operands reach (measured 2^-24.3 over 2e6 pairs)
    with np.errstate(over="ignore"):
        fl = (a * b).astype(np.float64)                                     # the correctly rounded float32 product, for comparison
    ok = nz & np.isfinite(fl) & (np.abs(exact) > 1e-30) & (np.abs(exact) < 1e30)
    rel32 = np.abs(fl[ok] - exact[ok]) / np.abs(exact[ok])
    assert rel32.max() <= 2.0 ** -24 and rel32.max() > 2.0 ** -24.1         # the float32 multiply's own rounding: bound 2^-24, reached
    rms6, rms32 = np.sqrt((rel6 ** 2).mean()), np.sqrt((rel32 ** 2).mean())
    assert rms6 < 0.3 * rms32                                               # on average a quarter of one float32 rounding (2^-27.4 vs 2^-25.2)
    # the two-piece modes: 3 terms of a 16-bit (bf16x3) / 22-bit (f16x3) operand split
    def two(v):                                                             # bf16x3's operands: hi + lo, 16 significand bits
        hi = bf16(v)
        return hi, bf16(v - hi)
    (a1, a2), (b1, b2) = two(a), two(b)
    k3 = a2.astype(np.float64) * b1 + a1.astype(np.float64) * b2 + a1.astype(np.float64) * b1
    assert 2.0 ** -18 < (np.abs(k3[nz] - exact[nz]) / np.abs(exact[nz])).max() < 2.0 ** -14
    m = (np.abs(a) > 2.0 ** -3) & (np.abs(a) < 1e4) & (np.abs(b) > 2.0 ** -3) & (np.abs(b) < 1e4)     # float16's comfortable range
    ah = a[m].astype(np.float16).astype(np.float32); al = (a[m] - ah).astype(np.float16).astype(np.float32)
    bh = b[m].astype(np.float16).astype(np.float32); bl = (b[m] - bh).astype(np.float16).astype(np.float32)
    f3 = al.astype(np.float64) * bh + ah.astype(np.float64) * bl + ah.astype(np.float64) * bh
    relf = np.abs(f3 - exact[m]) / np.abs(exact[m])
    assert 2.0 ** -24 < relf.max() < 2.0 ** -19
