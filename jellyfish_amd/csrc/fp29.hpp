// Unsaturated Montgomery arithmetic for the hot loops: L limbs of 29 bits in 32-bit registers, R' = 2^(29 L).
//
// Why: on gfx950 v_mad_u64_u32 issues at the full VALU rate, but it has no carry-in, so with saturated 32-bit limbs
// every limb product costs a second instruction (v_addc_co_u32 on the third accumulator word): 2 (2N^2 + N) = 272 issue
// slots for N = 8 before the shifts and the final conditional subtraction (~325 measured per product).  With 29-bit
// limbs a 64-bit accumulator absorbs a whole column of products (<= 2 L products of < 2^58) with no carry handling:
// 2 L^2 + L multiply-adds (171 for L = 9) plus two shift/mask instructions per column, and the result needs no final
// subtraction because the representation has >= 6 spare bits (values are kept "lazily" reduced, < 2^(29 L - 4)).
//
// This form is INTERNAL to kernels (SRS window table, bucket accumulators): data crossing the C ABI stays in arkworks'
// saturated Montgomery form with R = 2^(32 N); f29_from_fp / f29_to_fp convert (a few doublings / halvings: R'/R = 2^k).
#pragma once
#include <utility>

#include "fp.hpp"

namespace jf {

template <class C>
struct F29Params {
    static constexpr int L = (C::BITS + 3 + 28) / 29;
    static constexpr int SHIFT = 29 * L - 32 * C::N;  // R' / R = 2^SHIFT
    static constexpr uint32_t MASK = (1u << 29) - 1u;
    // limb i of p in radix 2^29
    JF_HD static constexpr uint32_t P(int i) {
        const int bit = 29 * i;
        const int w = bit / 32, s = bit % 32;
        uint64_t lo = w < C::N ? C::P(w) : 0u;
        uint64_t hi = (w + 1) < C::N ? C::P(w + 1) : 0u;
        return (uint32_t)(((lo | (hi << 32)) >> s) & MASK);
    }
    // -p^{-1} mod 2^29 (C::INV is -p^{-1} mod 2^32)
    static constexpr uint32_t INV = C::INV & MASK;
};


// acc >>= 29 on 32-bit halves (v_alignbit_b32 + v_lshrrev_b32; the 64-bit shift instruction is not full rate)
JF_HD uint64_t f29_shr29(uint64_t acc) {
    const uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t nlo = __builtin_amdgcn_alignbit(hi, lo, 29);
#else
    const uint32_t nlo = (lo >> 29) | (hi << 3);
#endif
    return ((uint64_t)(hi >> 29) << 32) | nlo;
}
// low 29 bits of x * inv: through the full-rate 32x32->64 multiply-add instead of v_mul_lo_u32
JF_HD uint32_t f29_mulmask(uint32_t x, uint32_t inv) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t t;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"(x), "s"(inv) : "vcc");
    return (uint32_t)t & ((1u << 29) - 1u);
#else
    return (x * inv) & ((1u << 29) - 1u);
#endif
}

template <class C>
struct F29 {
    static constexpr int L = F29Params<C>::L;
    uint32_t v[L];
};

// carry-propagate so that every limb but the top is < 2^29 (limbs may hold up to 32 bits on entry)
template <class C>
JF_HD F29<C> f29_normalize(const F29<C>& a) {
    constexpr int L = F29<C>::L;
    F29<C> r;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
        uint32_t t = a.v[i] + carry;  // < 2^32 as long as limbs < 2^32 - 8
        r.v[i] = t & F29Params<C>::MASK;
        carry = t >> 29;
    }
    r.v[L - 1] = a.v[L - 1] + carry;
    return r;
}

// lazy add: limb-wise, no carries (limbs grow by one bit)
template <class C>
JF_HD F29<C> f29_add_lazy(const F29<C>& a, const F29<C>& b) {
    F29<C> r;
#pragma unroll
    for (int i = 0; i < F29<C>::L; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}

// a + K p - b, normalised.  Requires b < K p (as integers) and normalised inputs; K p is added limb-wise with a
// borrow-proof bias: limb i of (K p) is taken from the radix-2^29 expansion of K p with 2^29 lent from the next limb.
template <class C, int K>
struct F29Kp {
    // limbs of K*p in radix 2^29 (K small), little-endian, computed with carries
    JF_HD static constexpr uint32_t limb(int i) {
        uint64_t carry = 0;
        uint32_t out = 0;
        for (int k = 0; k <= i; k++) {
            uint64_t t = (uint64_t)F29Params<C>::P(k) * K + carry;
            out = (uint32_t)(t & F29Params<C>::MASK);
            carry = t >> 29;
            if (k == F29Params<C>::L - 1) out = (uint32_t)t;  // top limb keeps the overflow
        }
        return out;
    }
    // biased limbs: b_i = limb_i + 2^29 (i < L-1) - 1 (i > 0): every b_i >= 2^29 - 1 > any normalised limb of b
    JF_HD static constexpr uint32_t biased(int i) {
        constexpr int L = F29Params<C>::L;
        uint32_t x = limb(i);
        if (i < L - 1) x += (1u << 29);
        if (i > 0) x -= 1u;
        return x;
    }
};

template <class C, int K>
JF_HD F29<C> f29_sub(const F29<C>& a, const F29<C>& b) {
    constexpr int L = F29<C>::L;
    F29<C> t;
#pragma unroll
    for (int i = 0; i < L; i++) t.v[i] = a.v[i] + F29Kp<C, K>::biased(i) - b.v[i];
    return f29_normalize<C>(t);
}

// a + K p - b - c - d in one pass, normalised.  Requires b + c + d < K p and normalised inputs.  Each limb borrows
// 2^31 (four units of the next limb) so the limb-wise differences never go negative.
template <class C, int K>
struct F29Kp4 {
    JF_HD static constexpr uint32_t biased(int i) {
        constexpr int L = F29Params<C>::L;
        uint32_t x = F29Kp<C, K>::limb(i);
        if (i < L - 1) x += (1u << 31);
        if (i > 0) x -= 4u;
        return x;
    }
};
template <class C, int K>
JF_HD F29<C> f29_sub3(const F29<C>& a, const F29<C>& b, const F29<C>& c, const F29<C>& d) {
    constexpr int L = F29<C>::L;
    F29<C> t;
#pragma unroll
    for (int i = 0; i < L; i++) t.v[i] = a.v[i] + F29Kp4<C, K>::biased(i) - b.v[i] - c.v[i] - d.v[i];
    return f29_normalize<C>(t);
}

// is a product output (normalised, < 2 p) congruent to 0, i.e. equal to 0 or to p ?
template <class C>
JF_HD bool f29_mulout_is_zero(const F29<C>& a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < F29<C>::L; i++) {
        z |= a.v[i];
        e |= a.v[i] ^ F29Params<C>::P(i);
    }
    return z == 0 || e == 0;
}

// re-slice 32-bit words holding a fully reduced value into 29-bit limbs (no Montgomery change).  On the device every
// limb is one v_alignbit_b32 / shift + one mask on 32-bit registers: written with 64-bit shifts the compiler wants the two
// source words in an aligned register pair and, for words that come straight from memory, fetches them a second time
// with an overlapping load (8 loads and 24 registers for a 64-byte table point instead of 4 and 16).
template <class C>
JF_HD F29<C> f29_unpack_bits(const Fp<C>& x) {
    constexpr int L = F29<C>::L;
    F29<C> r;
#pragma unroll
    for (int i = 0; i < L; i++) {
        const int bit = 29 * i, w = bit / 32, s = bit % 32;
        const uint32_t lo = w < C::N ? x.v[w] : 0u;
        const uint32_t hi = (w + 1) < C::N ? x.v[w + 1] : 0u;
        uint32_t t;
        if (s == 0) t = lo;
        else if (s + 29 <= 32 || (w + 1) >= C::N) t = lo >> s;
        else {
#if defined(__HIP_DEVICE_COMPILE__)
            t = __builtin_amdgcn_alignbit(hi, lo, s);
#else
            t = (lo >> s) | (hi << (32 - s));
#endif
        }
        r.v[i] = t & F29Params<C>::MASK;
    }
    return r;
}

template <class C>
JF_HD F29<C> f29_dbl_lazy(const F29<C>& a) {
    F29<C> r;
#pragma unroll
    for (int i = 0; i < F29<C>::L; i++) r.v[i] = a.v[i] << 1;
    return r;
}

// Montgomery product, product scanning, one 64-bit accumulator, no carry instructions.
// Inputs: limbs < 2^29 (one operand may have limbs < 2^30 for L <= 9), values < 2^(29 L - 4).
// Output: normalised limbs, value < a b / R' + p.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(JF_FP_MUL_PORTABLE)
// The multiply-adds of one column are ONE serial chain on the accumulator, written as inline asm: left to itself the
// compiler re-associates the column into many independent 64-bit partial sums and then spends one 64-bit add per
// product merging them (measured: more instructions than the saturated multiplier it replaces, and 180 VGPRs).
#define JF_MAD29_TXT(A, B) "v_mad_u64_u32 %0, vcc, %" #A ", %" #B ", %0\n\t"
JF_DEV void f29_mad(uint64_t& acc, uint32_t a, uint32_t b) {
    asm(JF_MAD29_TXT(1, 2) : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
JF_DEV void f29_mad_s(uint64_t& acc, uint32_t a, uint32_t b_uniform) {
    asm(JF_MAD29_TXT(1, 2) : "+v"(acc) : "v"(a), "s"(b_uniform) : "vcc");
}
// CNT pairs (a[J] b[I-J], m[J] p[I-J]), J = J0 .. J0+CNT-1, of column I
template <class C, int I, int J, int CNT>
JF_DEV void f29_col_pairs(uint64_t& acc, const uint32_t* a, const uint32_t* b, const uint32_t* m) {
    using PP = F29Params<C>;
    if constexpr (CNT >= 4) {
        asm(JF_MAD29_TXT(1, 2) JF_MAD29_TXT(3, 4) JF_MAD29_TXT(5, 6) JF_MAD29_TXT(7, 8) JF_MAD29_TXT(9, 10)
                JF_MAD29_TXT(11, 12) JF_MAD29_TXT(13, 14) JF_MAD29_TXT(15, 16)
            : "+v"(acc)
            : "v"(a[J]), "v"(b[I - J]), "v"(m[J]), "s"(PP::P(I - J)), "v"(a[J + 1]), "v"(b[I - J - 1]), "v"(m[J + 1]),
              "s"(PP::P(I - J - 1)), "v"(a[J + 2]), "v"(b[I - J - 2]), "v"(m[J + 2]), "s"(PP::P(I - J - 2)),
              "v"(a[J + 3]), "v"(b[I - J - 3]), "v"(m[J + 3]), "s"(PP::P(I - J - 3))
            : "vcc");
        f29_col_pairs<C, I, J + 4, CNT - 4>(acc, a, b, m);
    } else if constexpr (CNT >= 2) {
        asm(JF_MAD29_TXT(1, 2) JF_MAD29_TXT(3, 4) JF_MAD29_TXT(5, 6) JF_MAD29_TXT(7, 8)
            : "+v"(acc)
            : "v"(a[J]), "v"(b[I - J]), "v"(m[J]), "s"(PP::P(I - J)), "v"(a[J + 1]), "v"(b[I - J - 1]), "v"(m[J + 1]),
              "s"(PP::P(I - J - 1))
            : "vcc");
        f29_col_pairs<C, I, J + 2, CNT - 2>(acc, a, b, m);
    } else if constexpr (CNT == 1) {
        asm(JF_MAD29_TXT(1, 2) JF_MAD29_TXT(3, 4)
            : "+v"(acc)
            : "v"(a[J]), "v"(b[I - J]), "v"(m[J]), "s"(PP::P(I - J))
            : "vcc");
    }
}
template <class C, int I>
JF_DEV void f29_cols_low(uint64_t& acc, const uint32_t* a, const uint32_t* b, uint32_t* m) {
    using PP = F29Params<C>;
    if constexpr (I < PP::L) {
        f29_col_pairs<C, I, 0, I>(acc, a, b, m);
        f29_mad(acc, a[I], b[0]);
        m[I] = ((uint32_t)acc * PP::INV) & PP::MASK;  // v_mul_lo_u32: measured faster than a masked v_mad_u64_u32 here
        f29_mad_s(acc, m[I], PP::P(0));
        acc >>= 29;  // v_lshrrev_b64: measured faster than v_alignbit_b32 + v_lshrrev_b32
        f29_cols_low<C, I + 1>(acc, a, b, m);
    }
}
template <class C, int I>
JF_DEV void f29_cols_high(uint64_t& acc, const uint32_t* a, const uint32_t* b, const uint32_t* m, uint32_t* r) {
    using PP = F29Params<C>;
    if constexpr (I < 2 * PP::L - 1) {
        f29_col_pairs<C, I, I - PP::L + 1, 2 * PP::L - 1 - I>(acc, a, b, m);
        r[I - PP::L] = (uint32_t)acc & PP::MASK;
        acc >>= 29;  // v_lshrrev_b64: measured faster than v_alignbit_b32 + v_lshrrev_b32
        f29_cols_high<C, I + 1>(acc, a, b, m, r);
    }
}
template <class C>
JF_DEV F29<C> f29_mul(const F29<C>& a, const F29<C>& b) {
    constexpr int L = F29<C>::L;
    uint32_t m[L];
    F29<C> r;
    uint64_t acc = 0;
    f29_cols_low<C, 0>(acc, a.v, b.v, m);
    f29_cols_high<C, L>(acc, a.v, b.v, m, r.v);
    r.v[L - 1] = (uint32_t)acc;
    return r;
}
// ---- grouped multiply-adds ------------------------------------------------------------------------------
// N multiply-adds of one column in ONE asm statement.  Between two asm statements that hand a register on, the
// compiler's hazard recogniser (it cannot see inside) puts an s_nop; one statement per multiply-add measured 6 % slower
// in the bucket accumulation than statements of up to eight.
JF_DEV void f29_mad_vv(uint64_t& acc, uint32_t x0, uint32_t y0) {
    asm(JF_MAD29_TXT(1, 2) : "+v"(acc) : "v"(x0), "v"(y0) : "vcc");
}
JF_DEV void f29_mad_vv(uint64_t& acc, uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1) {
    asm(JF_MAD29_TXT(1, 3) JF_MAD29_TXT(2, 4) : "+v"(acc) : "v"(x0), "v"(x1), "v"(y0), "v"(y1) : "vcc");
}
JF_DEV void f29_mad_vv(uint64_t& acc, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t y0, uint32_t y1,
                       uint32_t y2, uint32_t y3) {
    asm(JF_MAD29_TXT(1, 5) JF_MAD29_TXT(2, 6) JF_MAD29_TXT(3, 7) JF_MAD29_TXT(4, 8)
        : "+v"(acc)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y0), "v"(y1), "v"(y2), "v"(y3)
        : "vcc");
}
JF_DEV void f29_mad_vv(uint64_t& acc, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4, uint32_t x5,
                       uint32_t x6, uint32_t x7, uint32_t y0, uint32_t y1, uint32_t y2, uint32_t y3, uint32_t y4,
                       uint32_t y5, uint32_t y6, uint32_t y7) {
    asm(JF_MAD29_TXT(1, 9) JF_MAD29_TXT(2, 10) JF_MAD29_TXT(3, 11) JF_MAD29_TXT(4, 12) JF_MAD29_TXT(5, 13)
            JF_MAD29_TXT(6, 14) JF_MAD29_TXT(7, 15) JF_MAD29_TXT(8, 16)
        : "+v"(acc)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(y0), "v"(y1), "v"(y2), "v"(y3),
          "v"(y4), "v"(y5), "v"(y6), "v"(y7)
        : "vcc");
}
// the same with uniform second factors (limbs of p: compile-time constants, SGPRs)
JF_DEV void f29_mad_vs(uint64_t& acc, uint32_t x0, uint32_t y0) {
    asm(JF_MAD29_TXT(1, 2) : "+v"(acc) : "v"(x0), "s"(y0) : "vcc");
}
JF_DEV void f29_mad_vs(uint64_t& acc, uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1) {
    asm(JF_MAD29_TXT(1, 3) JF_MAD29_TXT(2, 4) : "+v"(acc) : "v"(x0), "v"(x1), "s"(y0), "s"(y1) : "vcc");
}
JF_DEV void f29_mad_vs(uint64_t& acc, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t y0, uint32_t y1,
                       uint32_t y2, uint32_t y3) {
    asm(JF_MAD29_TXT(1, 5) JF_MAD29_TXT(2, 6) JF_MAD29_TXT(3, 7) JF_MAD29_TXT(4, 8)
        : "+v"(acc)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(y0), "s"(y1), "s"(y2), "s"(y3)
        : "vcc");
}
JF_DEV void f29_mad_vs(uint64_t& acc, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4, uint32_t x5,
                       uint32_t x6, uint32_t x7, uint32_t y0, uint32_t y1, uint32_t y2, uint32_t y3, uint32_t y4,
                       uint32_t y5, uint32_t y6, uint32_t y7) {
    asm(JF_MAD29_TXT(1, 9) JF_MAD29_TXT(2, 10) JF_MAD29_TXT(3, 11) JF_MAD29_TXT(4, 12) JF_MAD29_TXT(5, 13)
            JF_MAD29_TXT(6, 14) JF_MAD29_TXT(7, 15) JF_MAD29_TXT(8, 16)
        : "+v"(acc)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "s"(y0), "s"(y1), "s"(y2), "s"(y3),
          "s"(y4), "s"(y5), "s"(y6), "s"(y7)
        : "vcc");
}
// acc += sum over k in [0, CNT) of X(K0 + k) * Y(K0 + k), in statements of 8 / 4 / 2 / 1; X, Y: functors int -> uint32_t
template <int K0, int... K, class FX, class FY>
JF_DEV void f29_mad_seq_vv(uint64_t& acc, FX X, FY Y, std::integer_sequence<int, K...>) {
    f29_mad_vv(acc, X(std::integral_constant<int, K0 + K>{})..., Y(std::integral_constant<int, K0 + K>{})...);
}
template <int K0, int... K, class FX, class FY>
JF_DEV void f29_mad_seq_vs(uint64_t& acc, FX X, FY Y, std::integer_sequence<int, K...>) {
    f29_mad_vs(acc, X(std::integral_constant<int, K0 + K>{})..., Y(std::integral_constant<int, K0 + K>{})...);
}
template <bool UNIFORM, int K0, int CNT, class FX, class FY>
JF_DEV void f29_mad_run(uint64_t& acc, FX X, FY Y) {
    if constexpr (CNT > 0) {
        constexpr int step = CNT >= 8 ? 8 : CNT >= 4 ? 4 : CNT >= 2 ? 2 : 1;
        if constexpr (UNIFORM) f29_mad_seq_vs<K0>(acc, X, Y, std::make_integer_sequence<int, step>{});
        else f29_mad_seq_vv<K0>(acc, X, Y, std::make_integer_sequence<int, step>{});
        f29_mad_run<UNIFORM, K0 + step, CNT - step>(acc, X, Y);
    }
}
// column I of the reduction: m[J] p[I - J], J in [J0, JEND)
template <class C, int I, int J0, int JEND>
JF_DEV void f29_col_mp(uint64_t& acc, const uint32_t* m) {
    f29_mad_run<true, J0, JEND - J0>(
        acc, [&](auto k) { return m[decltype(k)::value]; },
        [&](auto k) { return F29Params<C>::P(I - decltype(k)::value); });
}

// a^2: the cross products a[J] a[I - J], J != I - J, come in equal pairs, so one multiply-add with the doubled operand
// d = 2 a (limbs < 2^30) covers two of them: 45 + 81 multiply-adds for L = 9 instead of 81 + 81
template <class C, int I>
JF_DEV void f29_sqr_col_ab(uint64_t& acc, const uint32_t* a, const uint32_t* d) {
    constexpr int L = F29Params<C>::L;
    constexpr int J0 = I < L ? 0 : I - L + 1;
    constexpr int JE = I / 2 + 1;   // J <= I - J
    f29_mad_run<false, J0, JE - J0>(
        acc, [&](auto k) { return a[decltype(k)::value]; },
        [&](auto k) {
            constexpr int J = decltype(k)::value;
            return J < I - J ? d[I - J] : a[J];
        });
}
template <class C, int I>
JF_DEV void f29_sqr_cols(uint64_t& acc, const uint32_t* a, const uint32_t* d, uint32_t* m, uint32_t* r) {
    using PP = F29Params<C>;
    constexpr int L = PP::L;
    if constexpr (I < L) {
        f29_sqr_col_ab<C, I>(acc, a, d);
        f29_col_mp<C, I, 0, I>(acc, m);
        m[I] = ((uint32_t)acc * PP::INV) & PP::MASK;
        f29_mad_s(acc, m[I], PP::P(0));
        acc >>= 29;
        f29_sqr_cols<C, I + 1>(acc, a, d, m, r);
    } else if constexpr (I < 2 * L - 1) {
        f29_sqr_col_ab<C, I>(acc, a, d);
        f29_col_mp<C, I, I - L + 1, L>(acc, m);
        r[I - L] = (uint32_t)acc & PP::MASK;
        acc >>= 29;
        f29_sqr_cols<C, I + 1>(acc, a, d, m, r);
    }
}
// Input: normalised limbs < 2^29 (the doubled copy is the operand with limbs < 2^30).
template <class C>
JF_DEV F29<C> f29_sqr(const F29<C>& a) {
    constexpr int L = F29<C>::L;
    uint32_t m[L], d[L];
#pragma unroll
    for (int i = 0; i < L; i++) d[i] = a.v[i] << 1;
    F29<C> r;
    uint64_t acc = 0;
    f29_sqr_cols<C, 0>(acc, a.v, d, m, r.v);
    r.v[L - 1] = (uint32_t)acc;
    return r;
}

// a1 b1 + a2 b2 with ONE Montgomery reduction (3 L^2 multiply-adds instead of 4 L^2).  Limbs: all four operands
// normalised (< 2^29), a column then holds at most 2 L products < 2^58 plus L of m p: < 2^63.3 for L <= 14.
// Output: normalised limbs, value < (a1 b1 + a2 b2) / R' + p.
template <class C, int I>
JF_DEV void f29_dot2_cols(uint64_t& acc, const uint32_t* a1, const uint32_t* b1, const uint32_t* a2, const uint32_t* b2,
                          uint32_t* m, uint32_t* r) {
    using PP = F29Params<C>;
    constexpr int L = PP::L;
    constexpr int J0 = I < L ? 0 : I - L + 1;
    constexpr int JE = I < L ? I + 1 : L;
    if constexpr (I < 2 * L - 1) {
        f29_mad_run<false, J0, JE - J0>(
            acc, [&](auto k) { return a1[decltype(k)::value]; }, [&](auto k) { return b1[I - decltype(k)::value]; });
        f29_mad_run<false, J0, JE - J0>(
            acc, [&](auto k) { return a2[decltype(k)::value]; }, [&](auto k) { return b2[I - decltype(k)::value]; });
        if constexpr (I < L) {
            f29_col_mp<C, I, 0, I>(acc, m);
            m[I] = ((uint32_t)acc * PP::INV) & PP::MASK;
            f29_mad_s(acc, m[I], PP::P(0));
        } else {
            f29_col_mp<C, I, I - L + 1, L>(acc, m);
            r[I - L] = (uint32_t)acc & PP::MASK;
        }
        acc >>= 29;
        f29_dot2_cols<C, I + 1>(acc, a1, b1, a2, b2, m, r);
    }
}
template <class C>
JF_DEV F29<C> f29_dot2(const F29<C>& a1, const F29<C>& b1, const F29<C>& a2, const F29<C>& b2) {
    constexpr int L = F29<C>::L;
    uint32_t m[L];
    F29<C> r;
    uint64_t acc = 0;
    f29_dot2_cols<C, 0>(acc, a1.v, b1.v, a2.v, b2.v, m, r.v);
    r.v[L - 1] = (uint32_t)acc;
    return r;
}
#else
template <class C>
JF_HD F29<C> f29_mul(const F29<C>& a, const F29<C>& b) {
    constexpr int L = F29<C>::L;
    using PP = F29Params<C>;
    uint32_t m[L];
    F29<C> r;
    uint64_t acc = 0;
    for (int k = 0; k < L; k++) {
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * PP::P(k - i);
        m[k] = ((uint32_t)acc * PP::INV) & PP::MASK;
        acc += (uint64_t)m[k] * PP::P(0);
        acc >>= 29;
    }
    for (int k = L; k < 2 * L - 1; k++) {
        for (int i = k - L + 1; i < L; i++) {
            acc += (uint64_t)a.v[i] * b.v[k - i];
            acc += (uint64_t)m[i] * PP::P(k - i);
        }
        r.v[k - L] = (uint32_t)acc & PP::MASK;
        acc >>= 29;
    }
    r.v[L - 1] = (uint32_t)acc;
    return r;
}
template <class C>
JF_HD F29<C> f29_sqr(const F29<C>& a) {
    return f29_mul<C>(a, a);
}
template <class C>
JF_HD F29<C> f29_dot2(const F29<C>& a1, const F29<C>& b1, const F29<C>& a2, const F29<C>& b2) {
    constexpr int L = F29<C>::L;
    using PP = F29Params<C>;
    uint32_t m[L];
    F29<C> r;
    uint64_t acc = 0;
    for (int k = 0; k < L; k++) {
        for (int i = 0; i <= k; i++) acc += (uint64_t)a1.v[i] * b1.v[k - i] + (uint64_t)a2.v[i] * b2.v[k - i];
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * PP::P(k - i);
        m[k] = ((uint32_t)acc * PP::INV) & PP::MASK;
        acc += (uint64_t)m[k] * PP::P(0);
        acc >>= 29;
    }
    for (int k = L; k < 2 * L - 1; k++) {
        for (int i = k - L + 1; i < L; i++) {
            acc += (uint64_t)a1.v[i] * b1.v[k - i] + (uint64_t)a2.v[i] * b2.v[k - i];
            acc += (uint64_t)m[i] * PP::P(k - i);
        }
        r.v[k - L] = (uint32_t)acc & PP::MASK;
        acc >>= 29;
    }
    r.v[L - 1] = (uint32_t)acc;
    return r;
}
#endif

// ---- conversions (cold) -------------------------------------------------------------------------------
// saturated Montgomery (x R) -> 29-bit Montgomery (x R'): SHIFT doublings mod p, then re-slice the bits
template <class C>
JF_HD F29<C> f29_from_fp(Fp<C> x) {
    constexpr int L = F29<C>::L;
    for (int k = 0; k < F29Params<C>::SHIFT; k++) x = fp_dbl<C>(x);
    F29<C> r;
#pragma unroll
    for (int i = 0; i < L; i++) {
        const int bit = 29 * i, w = bit / 32, s = bit % 32;
        uint64_t lo = w < C::N ? x.v[w] : 0u;
        uint64_t hi = (w + 1) < C::N ? x.v[w + 1] : 0u;
        r.v[i] = (uint32_t)(((lo | (hi << 32)) >> s) & F29Params<C>::MASK);
    }
    return r;
}

// the raw bits of a fully reduced (< p) value, re-sliced to saturated limbs (no Montgomery change)
template <class C>
JF_HD Fp<C> f29_pack_bits(const F29<C>& a) {
    constexpr int L = F29<C>::L;
    Fp<C> r;
#pragma unroll
    for (int w = 0; w < C::N; w++) {
        // word w holds bits [32w, 32w+32)
        const int i0 = (32 * w) / 29, s0 = (32 * w) % 29;
        uint64_t t = (uint64_t)a.v[i0] >> s0;
        if (i0 + 1 < L) t |= (uint64_t)a.v[i0 + 1] << (29 - s0);
        if (i0 + 2 < L) t |= (uint64_t)a.v[i0 + 2] << (58 - s0);
        r.v[w] = (uint32_t)t;
    }
    return r;
}

// lazily reduced (< 2^ROUNDS p, normalised) -> fully reduced < p, still in 29-bit limbs
template <class C, int ROUNDS = 4>
JF_HD F29<C> f29_reduce_full(const F29<C>& a) {
    constexpr int L = F29<C>::L;
    F29<C> cur = f29_normalize<C>(a);
#pragma unroll
    for (int sh = ROUNDS - 1; sh >= 0; sh--) {
        // t = cur - (p << sh), limb-wise with signed borrow
        F29<C> t;
        int64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            int64_t d = (int64_t)cur.v[i] - ((int64_t)F29Params<C>::P(i) << sh) + borrow;
            if (i < L - 1) {
                t.v[i] = (uint32_t)(d & F29Params<C>::MASK);
                borrow = d >> 29;
            } else {
                t.v[i] = (uint32_t)d;
                borrow = d >> 32;
            }
        }
        const bool neg = borrow < 0;
#pragma unroll
        for (int i = 0; i < L; i++) cur.v[i] = neg ? cur.v[i] : t.v[i];
    }
    return cur;
}

// 29-bit Montgomery (x R', lazily reduced) -> saturated Montgomery (x R), fully reduced: ONE product with the raw
// bits of R mod p (x R' * R / R' = x R, output < p + a p / R' < 2 p) and one conditional subtraction
template <class C>
JF_HD Fp<C> f29_to_fp(const F29<C>& a) {
    return f29_pack_bits<C>(f29_reduce_full<C, 1>(f29_mul<C>(a, f29_unpack_bits<C>(Fp<C>::one()))));
}
// the same by halvings (no product): reference form, kept for the field tests
template <class C>
JF_HD Fp<C> f29_to_fp_halvings(const F29<C>& a) {
    Fp<C> x = f29_pack_bits<C>(f29_reduce_full<C>(a));
    for (int k = 0; k < F29Params<C>::SHIFT; k++) {
        // x / 2 mod p
        uint32_t carry = 0;
        const uint32_t odd = 0u - (x.v[0] & 1u);
        Fp<C> t;
#pragma unroll
        for (int i = 0; i < C::N; i++) t.v[i] = jf_addc(x.v[i], C::P(i) & odd, carry);
#pragma unroll
        for (int i = 0; i < C::N - 1; i++) x.v[i] = (t.v[i] >> 1) | (t.v[i + 1] << 31);
        x.v[C::N - 1] = (t.v[C::N - 1] >> 1) | (carry << 31);
    }
    return x;
}

// x -> x 2^SHIFT mod p and back, on saturated values: changes the Montgomery factor R <-> R' without re-slicing
template <class C>
JF_HD Fp<C> fp_scale_up_r29(Fp<C> x) {
    for (int k = 0; k < F29Params<C>::SHIFT; k++) x = fp_dbl<C>(x);
    return x;
}
template <class C>
JF_HD Fp<C> fp_scale_down_r29(Fp<C> x) {
    for (int k = 0; k < F29Params<C>::SHIFT; k++) {
        uint32_t carry = 0;
        const uint32_t odd = 0u - (x.v[0] & 1u);
        Fp<C> t;
#pragma unroll
        for (int i = 0; i < C::N; i++) t.v[i] = jf_addc(x.v[i], C::P(i) & odd, carry);
#pragma unroll
        for (int i = 0; i < C::N - 1; i++) x.v[i] = (t.v[i] >> 1) | (t.v[i + 1] << 31);
        x.v[C::N - 1] = (t.v[C::N - 1] >> 1) | (carry << 31);
    }
    return x;
}

// x * t for a saturated, fully reduced x (x R) and a constant t kept in the 29-bit form with the factor R' (t R'):
// the product x t R comes back saturated and fully reduced -- the data keeps its Montgomery factor, the re-slicing costs
// ~50 instructions and the product runs on the carry-free multiplier (~280 instructions against ~320 of fp_mul).
// f29_mul's output is < p + x t R' / R' <= p + 2^256 p / 2^(29 L) < 2 p < 2^(32 N): it fits the words and one conditional
// subtraction reduces it.
template <class C>
JF_DEV Fp<C> fp_mul_const29(const Fp<C>& x, const F29<C>& t) {
    return fp_reduce_once<C>(f29_pack_bits<C>(f29_mul<C>(f29_unpack_bits<C>(x), t)));
}
// a saturated Montgomery constant (t R) -> that form (t R')
template <class C>
JF_HD F29<C> f29_const_from_fp(const Fp<C>& t) {
    return f29_unpack_bits<C>(fp_scale_up_r29<C>(t));
}

}  // namespace jf
