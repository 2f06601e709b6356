// G1 group law on unsaturated 29-bit limbs (fp29.hpp), lazily reduced: the representation the MSM accumulation writes
// (k_msm_accumulate29) and the one its tail stages and the group-element DFT (ecntt.hip) add in.
// Invariants of an element, in multiples of p (BN254 Fq is the tightest field: 2^257 = 10.58 p is the contraction limit
// of f29_mul): X < 7.66, Y < 3.66, ZZ, ZZZ < 1.66; all-zero limbs with inf set = identity.
#pragma once
#include "curve.hpp"
#include "fp29.hpp"

namespace jf {

template <class Cv>
struct XYZZ29 {
    using F = F29<typename Cv::Fq>;
    F X, Y, ZZ, ZZZ;
};

template <class Cv>
JF_DEV XYZZ<Cv> xyzz29_to_std(const XYZZ29<Cv>& a, bool inf) {
    using Q = typename Cv::Fq;
    if (inf) return XYZZ<Cv>::identity();
    XYZZ<Cv> r;
    r.X = f29_to_fp<Q>(a.X);
    r.Y = f29_to_fp<Q>(a.Y);
    r.ZZ = f29_to_fp<Q>(a.ZZ);
    r.ZZZ = f29_to_fp<Q>(a.ZZZ);
    return r;
}

template <class Cv>
struct P29 {
    XYZZ29<Cv> v;
    bool inf;
};

// saturated XYZZ (fully reduced) -> this representation
template <class Cv>
JF_DEV P29<Cv> p29_from_std(const XYZZ<Cv>& x) {
    using Q = typename Cv::Fq;
    P29<Cv> r;
    r.inf = x.is_inf();
    r.v.X = f29_from_fp<Q>(x.X);
    r.v.Y = f29_from_fp<Q>(x.Y);
    r.v.ZZ = f29_from_fp<Q>(x.ZZ);
    r.v.ZZZ = f29_from_fp<Q>(x.ZZZ);
    return r;
}

// cold: P + P, P - P (the general formulas divide by zero there), through the complete saturated addition
template <class Cv>
JF_COLD P29<Cv> p29_add_rare(P29<Cv> a, P29<Cv> b) {
    XYZZ<Cv> x = xyzz29_to_std<Cv>(a.v, a.inf), y = xyzz29_to_std<Cv>(b.v, b.inf);
    xyzz_add<Cv>(x, y);
    return p29_from_std<Cv>(x);
}

// acc += q (add-2008-s on XYZZ).  Bounds in multiples of p, products < 1.1 p for factors within the invariants:
// P, R = difference + 2p < 3.1; X3 = R^2 - PPP - 2Q + 6p < 7.1; Q - X3 + 8p < 9.1 (< 10.58, the contraction limit of
// f29_mul for BN254); Y3 = R (Q - X3 + 8p) + (2p - S1) PPP in one reduction < 1.2; ZZ3, ZZZ3 < 1.1.
template <class Cv>
JF_DEV void p29_add(P29<Cv>& acc, const P29<Cv>& q) {
    using Q = typename Cv::Fq;
    using F = F29<Q>;
    if (q.inf) return;
    if (acc.inf) {
        acc = q;
        return;
    }
    F zero29;
#pragma unroll
    for (int i = 0; i < F::L; i++) zero29.v[i] = 0;
    const F U1 = f29_mul<Q>(acc.v.X, q.v.ZZ);
    const F U2 = f29_mul<Q>(q.v.X, acc.v.ZZ);
    const F S1 = f29_mul<Q>(acc.v.Y, q.v.ZZZ);
    const F S2 = f29_mul<Q>(q.v.Y, acc.v.ZZZ);
    const F P = f29_sub<Q, 2>(U2, U1);
    const F PP = f29_sqr<Q>(P);
    if (f29_mulout_is_zero<Q>(PP)) {
        acc = p29_add_rare<Cv>(acc, q);
        return;
    }
    const F R = f29_sub<Q, 2>(S2, S1);
    const F PPP = f29_mul<Q>(P, PP);
    const F Qv = f29_mul<Q>(U1, PP);
    const F X3 = f29_sub3<Q, 6>(f29_sqr<Q>(R), PPP, Qv, Qv);
    const F Y3 = f29_dot2<Q>(R, f29_sub<Q, 8>(Qv, X3), f29_sub<Q, 2>(zero29, S1), PPP);
    acc.v.ZZ = f29_mul<Q>(f29_mul<Q>(acc.v.ZZ, q.v.ZZ), PP);
    acc.v.ZZZ = f29_mul<Q>(f29_mul<Q>(acc.v.ZZZ, q.v.ZZZ), PPP);
    acc.v.X = X3;
    acc.v.Y = Y3;
}

// 2 a (dbl-2008-s-1, curve coefficient a = 0).  U = 2Y < 7.4, V = U^2 < 1.4, W = U V, S = X V < 1.1, M = 3 X^2 < 4.1,
// X3 = M^2 - 2S + 4p < 5.1, Y3 = M (S - X3 + 6p) + (2p - W) Y < 1.3.  No point of order two exists (odd group order).
template <class Cv>
JF_DEV P29<Cv> p29_double(const P29<Cv>& a) {
    using Q = typename Cv::Fq;
    using F = F29<Q>;
    if (a.inf) return a;
    F zero29;
#pragma unroll
    for (int i = 0; i < F::L; i++) zero29.v[i] = 0;
    const F U = f29_normalize<Q>(f29_dbl_lazy<Q>(a.v.Y));
    const F V = f29_sqr<Q>(U);
    const F W = f29_mul<Q>(U, V);
    const F S = f29_mul<Q>(a.v.X, V);
    const F X2 = f29_sqr<Q>(a.v.X);
    const F M = f29_normalize<Q>(f29_add_lazy<Q>(X2, f29_dbl_lazy<Q>(X2)));
    P29<Cv> r;
    r.inf = false;
    r.v.X = f29_sub3<Q, 4>(f29_sqr<Q>(M), S, S, zero29);
    r.v.Y = f29_dot2<Q>(M, f29_sub<Q, 6>(S, r.v.X), f29_sub<Q, 2>(zero29, W), a.v.Y);
    r.v.ZZ = f29_mul<Q>(V, a.v.ZZ);
    r.v.ZZZ = f29_mul<Q>(W, a.v.ZZZ);
    return r;
}

// -a.  Y must be below K p: K = 1 for a point lifted from reduced coordinates, 2 for the output of p29_add / p29_double.
template <class Cv, int K>
JF_DEV P29<Cv> p29_neg(const P29<Cv>& a) {
    using Q = typename Cv::Fq;
    F29<Q> zero29;
#pragma unroll
    for (int i = 0; i < F29<Q>::L; i++) zero29.v[i] = 0;
    P29<Cv> r = a;
    r.v.Y = f29_sub<Q, K>(zero29, a.v.Y);
    return r;
}

// k a, k a canonical (non-Montgomery) integer of Fr::N words, a lifted from reduced coordinates: left-to-right double and
// add over the non-adjacent form of k (digits +-1 at the positions where 3k and k differ: one addition per three bits on
// average instead of one per two).  Out of line: callers are latency-bound kernels, one call per butterfly.
template <class Cv>
JF_COLD P29<Cv> p29_mul_scalar(P29<Cv> a, Fp<typename Cv::Fr> k) {
    constexpr int N = Cv::Fr::N;
    P29<Cv> r;
    r.inf = true;
#pragma unroll
    for (int i = 0; i < F29<typename Cv::Fq>::L; i++) r.v.X.v[i] = r.v.Y.v[i] = r.v.ZZ.v[i] = r.v.ZZZ.v[i] = 0;
    if (a.inf) return r;
    uint32_t h[N + 1], kk[N + 1];   // h = 3 k
    {
        uint32_t carry = 0, prev = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t two = (k.v[i] << 1) | prev;
            prev = k.v[i] >> 31;
            const uint64_t t = (uint64_t)k.v[i] + two + carry;
            h[i] = (uint32_t)t;
            carry = (uint32_t)(t >> 32);
            kk[i] = k.v[i];
        }
        h[N] = prev + carry;
        kk[N] = 0;
    }
    const P29<Cv> na = p29_neg<Cv, 1>(a);
    int top = -1;
    for (int i = N; i >= 0 && top < 0; i--) {
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j <= N; j++) w = (j == i) ? h[j] : w;
        if (w) top = 32 * i + 31 - __clz(w);
    }
    for (int bit = top; bit >= 1; bit--) {   // digit of weight 2^(bit - 1)
        r = p29_double<Cv>(r);
        uint32_t hw = 0, kw = 0;
#pragma unroll
        for (int j = 0; j <= N; j++) {
            hw = (j == (bit >> 5)) ? h[j] : hw;
            kw = (j == (bit >> 5)) ? kk[j] : kw;
        }
        const uint32_t hb = (hw >> (bit & 31)) & 1u, kb = (kw >> (bit & 31)) & 1u;
        if (hb != kb) p29_add<Cv>(r, hb ? a : na);
    }
    return r;
}

}  // namespace jf
