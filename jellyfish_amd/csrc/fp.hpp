// Montgomery prime-field arithmetic on 32-bit limbs for gfx950 VALU.
//
// Representation = arkworks' in-memory Fp (Montgomery form, R = 2^(32*N), little-endian
// limbs; reference pcs/src/univariate_kzg/mod.rs:970-979 hands over exactly `x.0.0`), so
// host buffers are consumed and produced without conversion.  All results are fully
// reduced (< p), as arkworks requires of stored values.
//
// The multiplier is v_mad_u64_u32 (32x32+64 -> 64); every field here has at least one
// spare top bit (p < 2^(32N-1)), so the CIOS "no-carry" variant needs only N+1 limbs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jf {

#define JF_DEV __device__ __forceinline__
#define JF_HD __host__ __device__ __forceinline__
#define JF_COLD __device__ __noinline__

// ---- field parameter packs -------------------------------------------------------------
struct Bn254Fr {
    static constexpr int N = 8;
    static constexpr uint32_t INV = 0xefffffffu;
    static constexpr int BITS = 254;
    JF_HD static constexpr uint32_t P(int i) {
        constexpr uint32_t p[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                   0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return p[i];
    }
    // R mod p
    JF_HD static constexpr uint32_t ONE(int i) {
        constexpr uint32_t v[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                   0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return v[i];
    }
    // R^2 mod p
    JF_HD static constexpr uint32_t R2(int i) {
        constexpr uint32_t v[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                   0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return v[i];
    }
};
struct Bn254Fq {
    static constexpr int N = 8;
    static constexpr uint32_t INV = 0xe4866389u;
    static constexpr int BITS = 254;
    JF_HD static constexpr uint32_t P(int i) {
        constexpr uint32_t p[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                   0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return p[i];
    }
    JF_HD static constexpr uint32_t ONE(int i) {
        constexpr uint32_t v[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                   0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return v[i];
    }
    JF_HD static constexpr uint32_t R2(int i) {
        constexpr uint32_t v[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                   0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return v[i];
    }
};
struct Bls381Fr {
    static constexpr int N = 8;
    static constexpr uint32_t INV = 0xffffffffu;
    static constexpr int BITS = 255;
    JF_HD static constexpr uint32_t P(int i) {
        constexpr uint32_t p[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                   0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return p[i];
    }
    JF_HD static constexpr uint32_t ONE(int i) {
        constexpr uint32_t v[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                   0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return v[i];
    }
    JF_HD static constexpr uint32_t R2(int i) {
        constexpr uint32_t v[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                   0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return v[i];
    }
};
struct Bls381Fq {
    static constexpr int N = 12;
    static constexpr uint32_t INV = 0xfffcfffdu;
    static constexpr int BITS = 381;
    JF_HD static constexpr uint32_t P(int i) {
        constexpr uint32_t p[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                    0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return p[i];
    }
    JF_HD static constexpr uint32_t ONE(int i) {
        constexpr uint32_t v[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                    0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return v[i];
    }
    JF_HD static constexpr uint32_t R2(int i) {
        constexpr uint32_t v[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                    0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return v[i];
    }
};

// ---- element type ----------------------------------------------------------------------
template <class C>
struct Fp {
    static constexpr int N = C::N;
    uint32_t v[C::N];

    JF_HD static Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = 0;
        return r;
    }
    JF_HD static Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = C::ONE(i);
        return r;
    }
    JF_HD static Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = C::R2(i);
        return r;
    }
    JF_HD bool is_zero() const {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < N; i++) t |= v[i];
        return t == 0;
    }
    JF_HD bool operator==(const Fp& o) const {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < N; i++) t |= v[i] ^ o.v[i];
        return t == 0;
    }
    JF_HD bool operator!=(const Fp& o) const { return !(*this == o); }
};

// Carry chains are written with clang's add/sub-with-carry builtins: they lower to one
// v_add(c)_co_u32 / v_sub(b)_co_u32 per limb.  (The u64 idiom `(uint64_t)a + b + c` compiles to
// 64-bit adds, shifts and moves: ~3x the instructions, measured as a third of the NTT's issue slots.)
JF_HD uint32_t jf_addc(uint32_t a, uint32_t b, uint32_t& carry) {
    unsigned co;
    uint32_t r = __builtin_addc(a, b, carry, &co);
    carry = co;
    return r;
}
JF_HD uint32_t jf_subb(uint32_t a, uint32_t b, uint32_t& borrow) {
    unsigned bo;
    uint32_t r = __builtin_subc(a, b, borrow, &bo);
    borrow = bo;
    return r;
}

// r = a - p if a >= p (a < 2p)
template <class C>
JF_HD Fp<C> fp_reduce_once(const Fp<C>& a) {
    constexpr int N = C::N;
    Fp<C> s;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) s.v[i] = jf_subb(a.v[i], C::P(i), borrow);
    Fp<C> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = borrow ? a.v[i] : s.v[i];
    return r;
}

template <class C>
JF_HD Fp<C> fp_add(const Fp<C>& a, const Fp<C>& b) {
    constexpr int N = C::N;
    Fp<C> t;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < N; i++) t.v[i] = jf_addc(a.v[i], b.v[i], carry);
    // p < 2^(32N-1) so a+b never overflows N limbs
    return fp_reduce_once<C>(t);
}

template <class C>
JF_HD Fp<C> fp_sub(const Fp<C>& a, const Fp<C>& b) {
    constexpr int N = C::N;
    Fp<C> t;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) t.v[i] = jf_subb(a.v[i], b.v[i], borrow);
    const uint32_t mask = 0u - borrow;
    uint32_t carry = 0;
    Fp<C> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = jf_addc(t.v[i], C::P(i) & mask, carry);
    return r;
}

template <class C>
JF_HD Fp<C> fp_neg(const Fp<C>& a) {
    constexpr int N = C::N;
    Fp<C> r;
    uint32_t borrow = 0;
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < N; i++) nz |= a.v[i];
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint32_t d = jf_subb(C::P(i), a.v[i], borrow);
        r.v[i] = nz ? d : 0u;
    }
    return r;
}

template <class C>
JF_HD Fp<C> fp_dbl(const Fp<C>& a) {
    constexpr int N = C::N;
    Fp<C> t;
    uint32_t hi = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        t.v[i] = (a.v[i] << 1) | hi;
        hi = a.v[i] >> 31;
    }
    return fp_reduce_once<C>(t);
}

// Portable CIOS Montgomery product (operand scanning).  Host code and the reference point for
// the device version below.
template <class C>
JF_HD Fp<C> fp_mul_cios(const Fp<C>& a, const Fp<C>& b) {
    constexpr int N = C::N;
    uint32_t t[N + 1];
#pragma unroll
    for (int i = 0; i <= N; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
        const uint32_t bi = b.v[i];
#pragma unroll
        for (int j = 0; j < N; j++) {
            c = (uint64_t)a.v[j] * bi + t[j] + c;
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        uint32_t tn = t[N] + (uint32_t)c;  // no-carry: fits 32 bits
        const uint32_t m = t[0] * C::INV;
        c = (uint64_t)m * C::P(0) + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            c = (uint64_t)m * C::P(j) + t[j] + c;
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += tn;
        t[N - 1] = (uint32_t)c;
        t[N] = (uint32_t)(c >> 32);
    }
    Fp<C> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = t[i];
    return fp_reduce_once<C>(r);
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(JF_FP_MUL_PORTABLE)
// 96-bit column accumulator (lo64, hi32) += a * b : v_mad_u64_u32 accumulates in place and hands
// its carry-out straight to v_addc_co_u32 through VCC -- one multiply-add and one add per limb
// product, no register moves (the C++ form costs a 64-bit add plus two v_mov per product).
JF_DEV void jf_mac(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(lo), "+v"(hi)
        : "v"(a), "v"(b)
        : "vcc");
}
// same with a wave-uniform (modulus) limb held in an SGPR
JF_DEV void jf_mac_s(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b_uniform) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(lo), "+v"(hi)
        : "v"(a), "s"(b_uniform)
        : "vcc");
}

#define JF_MAC_TXT(A, B) "v_mad_u64_u32 %0, vcc, %" #A ", %" #B ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
// K (a*b, m*p) pairs of one column in ONE asm statement: hipcc pads an s_nop after every asm
// statement that feeds the next one, so fewer, longer statements issue fewer bubbles.
template <class C, int I, int J, int CNT>
JF_DEV void jf_col_pairs(uint64_t& lo, uint32_t& hi, const uint32_t* a, const uint32_t* b, const uint32_t* m) {
    if constexpr (CNT >= 4) {
        asm(JF_MAC_TXT(2, 3) JF_MAC_TXT(4, 5) JF_MAC_TXT(6, 7) JF_MAC_TXT(8, 9) JF_MAC_TXT(10, 11) JF_MAC_TXT(12, 13)
                JF_MAC_TXT(14, 15) JF_MAC_TXT(16, 17)
            : "+v"(lo), "+v"(hi)
            : "v"(a[J]), "v"(b[I - J]), "v"(m[J]), "s"(C::P(I - J)), "v"(a[J + 1]), "v"(b[I - J - 1]), "v"(m[J + 1]),
              "s"(C::P(I - J - 1)), "v"(a[J + 2]), "v"(b[I - J - 2]), "v"(m[J + 2]), "s"(C::P(I - J - 2)),
              "v"(a[J + 3]), "v"(b[I - J - 3]), "v"(m[J + 3]), "s"(C::P(I - J - 3))
            : "vcc");
        jf_col_pairs<C, I, J + 4, CNT - 4>(lo, hi, a, b, m);
    } else if constexpr (CNT >= 2) {
        asm(JF_MAC_TXT(2, 3) JF_MAC_TXT(4, 5) JF_MAC_TXT(6, 7) JF_MAC_TXT(8, 9)
            : "+v"(lo), "+v"(hi)
            : "v"(a[J]), "v"(b[I - J]), "v"(m[J]), "s"(C::P(I - J)), "v"(a[J + 1]), "v"(b[I - J - 1]), "v"(m[J + 1]),
              "s"(C::P(I - J - 1))
            : "vcc");
        jf_col_pairs<C, I, J + 2, CNT - 2>(lo, hi, a, b, m);
    } else if constexpr (CNT == 1) {
        asm(JF_MAC_TXT(2, 3) JF_MAC_TXT(4, 5)
            : "+v"(lo), "+v"(hi)
            : "v"(a[J]), "v"(b[I - J]), "v"(m[J]), "s"(C::P(I - J))
            : "vcc");
    }
}

template <class C, int I>
JF_DEV void jf_fips_low(uint64_t& lo, uint32_t& hi, const uint32_t* a, const uint32_t* b, uint32_t* m) {
    if constexpr (I < C::N) {
        jf_col_pairs<C, I, 0, I>(lo, hi, a, b, m);
        jf_mac(lo, hi, a[I], b[0]);
        m[I] = (uint32_t)lo * C::INV;
        jf_mac_s(lo, hi, m[I], C::P(0));
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
        jf_fips_low<C, I + 1>(lo, hi, a, b, m);
    }
}
template <class C, int I>
JF_DEV void jf_fips_high(uint64_t& lo, uint32_t& hi, const uint32_t* a, const uint32_t* b, const uint32_t* m,
                         uint32_t* r) {
    if constexpr (I < 2 * C::N) {
        jf_col_pairs<C, I, I - C::N + 1, 2 * C::N - 1 - I>(lo, hi, a, b, m);
        r[I - C::N] = (uint32_t)lo;
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
        jf_fips_high<C, I + 1>(lo, hi, a, b, m, r);
    }
}

// Montgomery product, finely integrated product scanning (FIPS): column i of a*b + m*p is summed
// in one 96-bit accumulator, m[i] is chosen to clear its low word, the accumulator shifts down.
template <class C>
JF_DEV Fp<C> fp_mul(const Fp<C>& a, const Fp<C>& b) {
    uint32_t m[C::N];
    Fp<C> r;
    uint64_t lo = 0;
    uint32_t hi = 0;
    jf_fips_low<C, 0>(lo, hi, a.v, b.v, m);
    jf_fips_high<C, C::N>(lo, hi, a.v, b.v, m, r.v);
    return fp_reduce_once<C>(r);
}
#else
template <class C>
JF_HD Fp<C> fp_mul(const Fp<C>& a, const Fp<C>& b) {
    return fp_mul_cios<C>(a, b);
}
#endif

// equality of two fully reduced values
template <class C>
JF_HD bool fp_eq(const Fp<C>& a, const Fp<C>& b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) d |= a.v[i] ^ b.v[i];
    return d == 0;
}

template <class C>
JF_HD Fp<C> fp_sqr(const Fp<C>& a) {
    return fp_mul<C>(a, a);
}

// Montgomery reduction of a single N-limb value: a * R^{-1} mod p  (Montgomery -> canonical)
template <class C>
JF_HD Fp<C> fp_from_mont(const Fp<C>& a) {
    constexpr int N = C::N;
    uint32_t t[N + 1];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = a.v[i];
    t[N] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const uint32_t m = t[0] * C::INV;
        uint64_t c = (uint64_t)m * C::P(0) + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            c = (uint64_t)m * C::P(j) + t[j] + c;
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        t[N - 1] = (uint32_t)c;
    }
    Fp<C> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = t[i];
    return fp_reduce_once<C>(r);
}

template <class C>
JF_HD Fp<C> fp_to_mont(const Fp<C>& a) {
    return fp_mul<C>(a, Fp<C>::r2());
}

// limb i of p - 2, with borrow propagation (BLS12-381 Fr has low limb 1)
template <class C>
JF_HD constexpr uint32_t pm2_limb(int i) {
    uint32_t borrow = 2, r = 0;
    for (int k = 0; k <= i; k++) {
        uint64_t d = (uint64_t)C::P(k) - borrow;
        r = (uint32_t)d;
        borrow = (uint32_t)(d >> 32) & 1u;
    }
    return r;
}

// a^(p-2) (Fermat); inv(0) = 0.  Exponent bits come from the constexpr modulus.
// Out of line: it is large and only ever on cold, latency-bound paths.
template <class C>
__device__ __noinline__ Fp<C> fp_inv(const Fp<C>& a) {
    constexpr int N = C::N;
    Fp<C> acc = Fp<C>::one();
    for (int i = N - 1; i >= 0; i--) {
        uint32_t e = 0;
#pragma unroll
        for (int j = 0; j < N; j++) e = (j == i) ? pm2_limb<C>(j) : e;
        for (int b = 31; b >= 0; b--) {
            acc = fp_sqr<C>(acc);
            if ((e >> b) & 1u) acc = fp_mul<C>(acc, a);
        }
    }
    return acc;
}

// Binary extended Euclid on the raw integers (HAC 14.61 shape): ~2*bits iterations of shifts and
// subtractions instead of ~1.5*bits Montgomery products.  Round 1's normalisation; the kernels now call
// fp_inv_safegcd below (7x faster on one lane, 40x in throughput: the branches here diverge across a wave), this one
// stays as a second opinion on the test surface (jfhip_field_op 13).  Input/output in Montgomery form:
// inv(aR) = a^-1 R^-1, lifted back with two multiplications by R^2.  inv(0) = 0.
template <class C>
__device__ __noinline__ Fp<C> fp_inv_binary(const Fp<C>& a) {
    constexpr int N = C::N;
    if (a.is_zero()) return a;
    uint32_t u[N], v[N], x1[N], x2[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        u[i] = a.v[i];
        v[i] = C::P(i);
        x1[i] = (i == 0) ? 1u : 0u;
        x2[i] = 0u;
    }
    auto is_one = [](const uint32_t* w) {
        uint32_t t = w[0] ^ 1u;
#pragma unroll
        for (int i = 1; i < N; i++) t |= w[i];
        return t == 0;
    };
    auto shr1 = [](uint32_t* w, uint32_t top) {
#pragma unroll
        for (int i = 0; i < N - 1; i++) w[i] = (w[i] >> 1) | (w[i + 1] << 31);
        w[N - 1] = (w[N - 1] >> 1) | (top << 31);
    };
    auto half_mod = [&](uint32_t* x) {  // x/2 mod p
        uint32_t carry = 0;
        if (x[0] & 1u) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                uint64_t t = (uint64_t)x[i] + C::P(i) + carry;
                x[i] = (uint32_t)t;
                carry = (uint32_t)(t >> 32);
            }
        }
        shr1(x, carry);
    };
    auto geq = [](const uint32_t* p1, const uint32_t* p2) {
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t d = (uint64_t)p1[i] - p2[i] - borrow;
            borrow = (uint32_t)(d >> 32) & 1u;
        }
        return borrow == 0;
    };
    auto sub_raw = [](uint32_t* p1, const uint32_t* p2) {  // p1 -= p2
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t d = (uint64_t)p1[i] - p2[i] - borrow;
            p1[i] = (uint32_t)d;
            borrow = (uint32_t)(d >> 32) & 1u;
        }
        return borrow;
    };
    auto sub_mod = [&](uint32_t* p1, const uint32_t* p2) {  // p1 = p1 - p2 mod p
        if (sub_raw(p1, p2)) {
            uint32_t carry = 0;
#pragma unroll
            for (int i = 0; i < N; i++) {
                uint64_t t = (uint64_t)p1[i] + C::P(i) + carry;
                p1[i] = (uint32_t)t;
                carry = (uint32_t)(t >> 32);
            }
        }
    };
    // bounded: each pass of the outer loop strips at least one bit of u or v
    for (int guard = 0; guard < 4 * 32 * N && !is_one(u) && !is_one(v); guard++) {
        while (!(u[0] & 1u)) {
            shr1(u, 0);
            half_mod(x1);
        }
        while (!(v[0] & 1u)) {
            shr1(v, 0);
            half_mod(x2);
        }
        if (geq(u, v)) {
            sub_raw(u, v);
            sub_mod(x1, x2);
        } else {
            sub_raw(v, u);
            sub_mod(x2, x1);
        }
    }
    Fp<C> r;
    const bool pick_u = is_one(u);
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = pick_u ? x1[i] : x2[i];
    r = fp_mul<C>(r, Fp<C>::r2());
    return fp_mul<C>(r, Fp<C>::r2());
}

// ---- inversion by batched division steps (Bernstein-Yang "safegcd", the shape of 30-bit signed-limb implementations) ---
// 30 division steps at a time are decided on the low 32 bits of (f, g) alone and collected in a 2 x 2 integer matrix
// T = [[u, v], [q, r]] with 2^30 (f', g') = T (f, g); the matrix is then applied once to the full-width (f, g) (exact
// division by 2^30) and to (d, e) modulo p (a multiple of p makes the low 30 bits vanish), which keep d a = f, e a = g
// (mod p).  g reaches 0 after at most (49 bits + 57) / 17 steps with f = +-1, so a^-1 = +-d.  About 12 instructions per
// step on 32-bit registers plus ~250 per batch, against ~70 per bit stripped for the binary Euclid above, and no
// data-dependent branch inside a batch (measured, scripts/inv_bench.py: one BN254 Fq inversion on one lane 0.15 ->
// 0.02 ms, 2^18 of them 0.85 -> 0.02 ms).  Values are signed, 30 bits per limb, the top limb carrying the sign.
template <class C>
struct Inv30 {
    static constexpr int BITS = 32 * C::N;
    static constexpr int NL = (BITS + 2 + 29) / 30;            // room for (-2p, p) and |f|, |g| <= 2^BITS
    static constexpr int MAX_BATCHES = ((49 * BITS + 57) / 17 + 29) / 30 + 1;
    JF_HD static constexpr int32_t plimb(int i) {              // limb i of p, 30 bits each
        const int bit = 30 * i, w = bit / 32, sh = bit % 32;
        uint64_t lo = w < C::N ? C::P(w) : 0u;
        uint64_t hi = (w + 1) < C::N ? C::P(w + 1) : 0u;
        return (int32_t)(((lo | (hi << 32)) >> sh) & 0x3fffffffu);
    }
    JF_HD static constexpr uint32_t pinv30() {                 // p^-1 mod 2^30 (Newton on the low word, p odd)
        uint32_t p0 = C::P(0), x = p0;                         // 3 correct bits
        for (int k = 0; k < 5; k++) x *= 2u - p0 * x;
        return x & 0x3fffffffu;
    }
};

template <class C>
__device__ __noinline__ Fp<C> fp_inv_safegcd(const Fp<C>& a) {
    using I = Inv30<C>;
    constexpr int N = C::N, NL = I::NL;
    constexpr int32_t M30 = 0x3fffffff;
    if (a.is_zero()) return a;
    int32_t f[NL], g[NL], d[NL], e[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = 30 * i, w = bit / 32, sh = bit % 32;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < N; k++) {   // static indices only
            lo = (k == w) ? a.v[k] : lo;
            hi = (k == w + 1) ? a.v[k] : hi;
        }
        const uint64_t two = (uint64_t)lo | ((uint64_t)hi << 32);
        f[i] = I::plimb(i);
        g[i] = (int32_t)((two >> sh) & 0x3fffffffu);
        d[i] = 0;
        e[i] = (i == 0) ? 1 : 0;
    }
    int32_t delta = 1;
    for (int batch = 0; batch < I::MAX_BATCHES; batch++) {
        uint32_t gz = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) gz |= (uint32_t)g[i];
        if (gz == 0) break;
        // 30 division steps on the low words
        uint32_t f0 = (uint32_t)f[0] | ((uint32_t)f[1] << 30), g0 = (uint32_t)g[0] | ((uint32_t)g[1] << 30);
        int32_t u = 1, v = 0, q = 0, r = 1;
        for (int k = 0; k < 30; k++) {
            const bool odd = g0 & 1u;
            const bool swap = odd && delta > 0;
            const uint32_t addf = odd ? (swap ? 0u - f0 : f0) : 0u;
            const int32_t addu = odd ? (swap ? -u : u) : 0, addv = odd ? (swap ? -v : v) : 0;
            const uint32_t nf = swap ? g0 : f0;
            const int32_t nu = swap ? q : u, nv = swap ? r : v;
            g0 = (g0 + addf) >> 1;
            q += addu;
            r += addv;
            f0 = nf;
            u = 2 * nu;
            v = 2 * nv;
            delta = swap ? 1 - delta : 1 + delta;
        }
        // (f, g) <- T (f, g) / 2^30, exactly
        {
            int64_t cf = (int64_t)u * f[0] + (int64_t)v * g[0];
            int64_t cg = (int64_t)q * f[0] + (int64_t)r * g[0];
            cf >>= 30;
            cg >>= 30;
#pragma unroll
            for (int i = 1; i < NL; i++) {
                cf += (int64_t)u * f[i] + (int64_t)v * g[i];
                cg += (int64_t)q * f[i] + (int64_t)r * g[i];
                f[i - 1] = (int32_t)cf & M30;
                g[i - 1] = (int32_t)cg & M30;
                cf >>= 30;
                cg >>= 30;
            }
            f[NL - 1] = (int32_t)cf;
            g[NL - 1] = (int32_t)cg;
        }
        // (d, e) <- T (d, e) / 2^30 mod p; d, e stay in (-2p, p)
        {
            const int32_t sd = d[NL - 1] >> 31, se = e[NL - 1] >> 31;
            int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
            int64_t cd = (int64_t)u * d[0] + (int64_t)v * e[0];
            int64_t ce = (int64_t)q * d[0] + (int64_t)r * e[0];
            md -= (int32_t)((I::pinv30() * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
            me -= (int32_t)((I::pinv30() * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
            cd += (int64_t)I::plimb(0) * md;
            ce += (int64_t)I::plimb(0) * me;
            cd >>= 30;
            ce >>= 30;
#pragma unroll
            for (int i = 1; i < NL; i++) {
                cd += (int64_t)u * d[i] + (int64_t)v * e[i] + (int64_t)I::plimb(i) * md;
                ce += (int64_t)q * d[i] + (int64_t)r * e[i] + (int64_t)I::plimb(i) * me;
                d[i - 1] = (int32_t)cd & M30;
                e[i - 1] = (int32_t)ce & M30;
                cd >>= 30;
                ce >>= 30;
            }
            d[NL - 1] = (int32_t)cd;
            e[NL - 1] = (int32_t)ce;
        }
    }
    // f = +-1: a^-1 = d f.  Bring it from (-2p, 2p) into [0, p).
    const bool fneg = f[NL - 1] < 0;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        c += fneg ? -(int64_t)d[i] : (int64_t)d[i];
        d[i] = (i < NL - 1) ? ((int32_t)c & M30) : (int32_t)c;
        if (i < NL - 1) c >>= 30;
    }
    for (int round = 0; round < 2; round++) {   // + p while negative
        if (d[NL - 1] >= 0) break;
        c = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            c += (int64_t)d[i] + I::plimb(i);
            d[i] = (i < NL - 1) ? ((int32_t)c & M30) : (int32_t)c;
            if (i < NL - 1) c >>= 30;
        }
    }
    for (int round = 0; round < 2; round++) {   // - p while >= p
        int32_t t[NL];
        c = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            c += (int64_t)d[i] - I::plimb(i);
            t[i] = (i < NL - 1) ? ((int32_t)c & M30) : (int32_t)c;
            if (i < NL - 1) c >>= 30;
        }
        if (t[NL - 1] < 0) break;
#pragma unroll
        for (int i = 0; i < NL; i++) d[i] = t[i];
    }
    Fp<C> res;
#pragma unroll
    for (int w = 0; w < N; w++) {   // word w = bits [32 w, 32 w + 32)
        const int i0 = (32 * w) / 30, s0 = (32 * w) % 30;
        uint64_t t = (uint64_t)(uint32_t)d[i0] >> s0;
        if (i0 + 1 < NL) t |= (uint64_t)(uint32_t)d[i0 + 1] << (30 - s0);
        if (i0 + 2 < NL) t |= (uint64_t)(uint32_t)d[i0 + 2] << (60 - s0);
        res.v[w] = (uint32_t)t;
    }
    res = fp_mul<C>(res, Fp<C>::r2());
    return fp_mul<C>(res, Fp<C>::r2());
}

// small-exponent power
template <class C>
JF_HD Fp<C> fp_pow_u64(Fp<C> base, uint64_t e) {
    Fp<C> acc = Fp<C>::one();
    while (e) {
        if (e & 1) acc = fp_mul<C>(acc, base);
        base = fp_sqr<C>(base);
        e >>= 1;
    }
    return acc;
}

// ---- global-memory access: N limbs as 16-byte vectors -----------------------------------
template <class C>
JF_DEV Fp<C> fp_load(const uint32_t* __restrict__ p) {
    Fp<C> r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < C::N / 4; i++) {
        uint4 t = q[i];
        r.v[4 * i] = t.x;
        r.v[4 * i + 1] = t.y;
        r.v[4 * i + 2] = t.z;
        r.v[4 * i + 3] = t.w;
    }
    return r;
}
template <class C>
JF_DEV void fp_store(uint32_t* __restrict__ p, const Fp<C>& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < C::N / 4; i++) q[i] = make_uint4(a.v[4 * i], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]);
}

}  // namespace jf
