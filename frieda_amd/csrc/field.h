// field.h — M31 / CM31 / QM31 arithmetic shared by host code and gfx950 kernels.
//
// Follows stwo-prover@19d12d7 core/fields/{m31,cm31,qm31}.rs (the field types frieda re-exports at
// /root/reference/src/lib.rs:14 and uses at src/proof.rs:6,25).  All values canonical in [0, P).
// 32-bit integer arithmetic only: the product is one 32x32->64 multiply (v_mad_u64_u32 / v_mul_hi+lo on
// CDNA4) followed by a shift-and-add fold of the Mersenne modulus; conditional subtractions are written as
// unsigned min() so that they compile to v_min_u32 instead of compare+select.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FR_HD __host__ __device__ __forceinline__
#else
#define FR_HD inline
#endif

namespace frieda {

constexpr uint32_t P31 = 0x7fffffffu;

FR_HD uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

FR_HD uint32_t m31_add(uint32_t a, uint32_t b) {
    uint32_t s = a + b;  // <= 2P-2 < 2^32
    return umin32(s, s - P31);
}
FR_HD uint32_t m31_sub(uint32_t a, uint32_t b) {
    uint32_t d = a - b;  // wraps when a < b; then d + P is the canonical value and is the smaller one
    return umin32(d, d + P31);
}
FR_HD uint32_t m31_neg(uint32_t a) { return m31_sub(0u, a); }
FR_HD uint32_t m31_mul(uint32_t a, uint32_t b) {
    uint64_t p = (uint64_t)a * b;
    uint32_t hi = (uint32_t)(p >> 31), lo = (uint32_t)p & P31;  // p = hi*2^31 + lo == hi + lo (mod P)
    uint32_t s = hi + lo;
    return umin32(s, s - P31);
}
FR_HD uint32_t m31_sqr(uint32_t a) { return m31_mul(a, a); }
// reduce a u32 < 2P (Blake2sChannel::draw_base_felts)
FR_HD uint32_t m31_reduce_2p(uint32_t v) { return umin32(v, v - P31); }

FR_HD uint32_t m31_pow(uint32_t a, uint32_t e) {
    uint32_t r = 1;
    while (e) {
        if (e & 1) r = m31_mul(r, a);
        a = m31_sqr(a);
        e >>= 1;
    }
    return r;
}
// a^(P-2) by a fixed addition chain: 2^31 - 3 = (2^29 - 1) * 4 + 1  ->  37 multiplications
FR_HD uint32_t m31_inv(uint32_t a) {
    auto sqn = [](uint32_t v, int n) {
        for (int i = 0; i < n; i++) v = m31_sqr(v);
        return v;
    };
    uint32_t t1 = m31_mul(m31_sqr(a), a);     // 2^2 - 1
    uint32_t t2 = m31_mul(sqn(t1, 2), t1);    // 2^4 - 1
    uint32_t t3 = m31_mul(sqn(t2, 4), t2);    // 2^8 - 1
    uint32_t t4 = m31_mul(sqn(t3, 8), t3);    // 2^16 - 1
    uint32_t t5 = m31_mul(sqn(t4, 8), t3);    // 2^24 - 1
    uint32_t t6 = m31_mul(sqn(t5, 4), t2);    // 2^28 - 1
    uint32_t t7 = m31_mul(m31_sqr(t6), a);    // 2^29 - 1
    return m31_mul(sqn(t7, 2), a);            // (2^29 - 1) * 4 + 1 = 2^31 - 3
}

struct CM31 {
    uint32_t a, b;  // a + b i
};
struct QM31 {
    uint32_t a, b, c, d;  // (a + b i) + (c + d i) u,  u^2 = 2 + i
};

FR_HD CM31 cm_add(CM31 x, CM31 y) { return {m31_add(x.a, y.a), m31_add(x.b, y.b)}; }
FR_HD CM31 cm_sub(CM31 x, CM31 y) { return {m31_sub(x.a, y.a), m31_sub(x.b, y.b)}; }
FR_HD CM31 cm_mul(CM31 x, CM31 y) {
    return {m31_sub(m31_mul(x.a, y.a), m31_mul(x.b, y.b)), m31_add(m31_mul(x.a, y.b), m31_mul(x.b, y.a))};
}
// multiply by R = 2 + i: (a + bi)(2 + i) = (2a - b) + (a + 2b) i
FR_HD CM31 cm_mul_r(CM31 x) {
    return {m31_sub(m31_add(x.a, x.a), x.b), m31_add(x.a, m31_add(x.b, x.b))};
}

FR_HD QM31 qm_add(QM31 x, QM31 y) { return {m31_add(x.a, y.a), m31_add(x.b, y.b), m31_add(x.c, y.c), m31_add(x.d, y.d)}; }
FR_HD QM31 qm_sub(QM31 x, QM31 y) { return {m31_sub(x.a, y.a), m31_sub(x.b, y.b), m31_sub(x.c, y.c), m31_sub(x.d, y.d)}; }
FR_HD QM31 qm_scale(QM31 x, uint32_t s) { return {m31_mul(x.a, s), m31_mul(x.b, s), m31_mul(x.c, s), m31_mul(x.d, s)}; }
FR_HD QM31 qm_mul(QM31 x, QM31 y) {
    CM31 x0{x.a, x.b}, x1{x.c, x.d}, y0{y.a, y.b}, y1{y.c, y.d};
    CM31 lo = cm_add(cm_mul(x0, y0), cm_mul_r(cm_mul(x1, y1)));
    CM31 hi = cm_add(cm_mul(x0, y1), cm_mul(x1, y0));
    return {lo.a, lo.b, hi.a, hi.b};
}
// Multiplication by a fixed s in QM31 is M31-linear: a 4x4 matrix on the coordinates (row = output coordinate).  With
// R = u^2 = 2 + i:  (A + B u)(C + D u) = AC + R BD + (AD + BC) u.
struct QM31Mat {
    uint32_t m[4][4];
};
FR_HD QM31Mat qm_matrix(QM31 s) {
    const uint32_t nb = m31_neg(s.b), nd = m31_neg(s.d);
    const uint32_t t = m31_sub(m31_add(s.c, s.c), s.d);  // Re(R B) = 2c - d
    const uint32_t u = m31_add(m31_add(s.d, s.d), s.c);  // Im(R B) = c + 2d
    return {{{s.a, nb, t, m31_neg(u)}, {s.b, s.a, u, t}, {s.c, nd, s.a, nb}, {s.d, s.c, s.b, s.a}}};
}
// any 64-bit value mod P, canonical: v = hi 2^32 + lo == 2 hi + lo (2^31 == 1), a 34-bit number, folded once more
FR_HD uint32_t m31_reduce64(uint64_t v) {
    const uint64_t w = (uint64_t)(uint32_t)(v >> 32) * 2u + (uint32_t)v;
    const uint32_t s = ((uint32_t)w & P31) + (uint32_t)(w >> 31);  // <= P + 7
    return umin32(s, s - P31);
}
// FRI fold of one pair of QM31 values: (x + y) + s * ((x - y) * it), `sm` = qm_matrix(s).  Each output coordinate is a sum of
// four 62-bit products plus a 32-bit term — it fits 64 bits exactly (4 (P-1)^2 + 2P < 2^64) — reduced once: 20 multiply-adds
// and 8 reductions instead of the 20 multiplications, 20 reductions and ~30 modular add/subs of the textbook sequence.
FR_HD QM31 qm_fold_pair(QM31 x, QM31 y, uint32_t it, const QM31Mat& sm) {
    const uint32_t e[4] = {m31_reduce64((uint64_t)(x.a + (P31 - y.a)) * it), m31_reduce64((uint64_t)(x.b + (P31 - y.b)) * it),
                           m31_reduce64((uint64_t)(x.c + (P31 - y.c)) * it), m31_reduce64((uint64_t)(x.d + (P31 - y.d)) * it)};
    const uint32_t f0[4] = {x.a + y.a, x.b + y.b, x.c + y.c, x.d + y.d};  // < 2P, unreduced
    uint32_t r[4];
    for (int k = 0; k < 4; k++) {
        uint64_t acc = f0[k];
        for (int j = 0; j < 4; j++) acc += (uint64_t)sm.m[k][j] * e[j];
        r[k] = m31_reduce64(acc);
    }
    return {r[0], r[1], r[2], r[3]};
}

FR_HD bool qm_eq(QM31 x, QM31 y) { return x.a == y.a && x.b == y.b && x.c == y.c && x.d == y.d; }
FR_HD bool qm_is_zero(QM31 x) { return (x.a | x.b | x.c | x.d) == 0; }

// circle group over M31 (stwo core/circle.rs); generator of the order-2^31 group
struct CPoint {
    uint32_t x, y;
};
constexpr uint32_t CIRCLE_GEN_X = 2u, CIRCLE_GEN_Y = 1268011823u;
FR_HD CPoint cp_add(CPoint p, CPoint q) {
    return {m31_sub(m31_mul(p.x, q.x), m31_mul(p.y, q.y)), m31_add(m31_mul(p.x, q.y), m31_mul(p.y, q.x))};
}
FR_HD CPoint cp_double(CPoint p) {
    uint32_t xx = m31_sqr(p.x), xy = m31_mul(p.x, p.y);
    return {m31_sub(m31_add(xx, xx), 1u), m31_add(xy, xy)};  // (2x^2 - 1, 2xy) on the unit circle
}
FR_HD uint32_t double_x(uint32_t x) {
    uint32_t xx = m31_sqr(x);
    return m31_sub(m31_add(xx, xx), 1u);
}

FR_HD uint32_t bit_reverse(uint32_t i, uint32_t log_size) {
    if (log_size == 0) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(i) >> (32 - log_size);
#else
    uint32_t r = 0;
    for (uint32_t b = 0; b < log_size; b++) r |= ((i >> b) & 1u) << (log_size - 1 - b);
    return r;
#endif
}

}  // namespace frieda
