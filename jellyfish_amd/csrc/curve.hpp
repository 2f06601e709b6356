// Short-Weierstrass G1 (a = 0) group arithmetic for the Pippenger MSM kernels.
//
// Bucket accumulators live in extended Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity <=> ZZ = 0): the mixed add of an affine SRS
// point costs 8M + 2S with no inversion.  Every formula handles the exceptional inputs
// (identity operand, P + P, P + (-P)) so results are exact for any scalars / bases, which
// is what makes the outputs bit-identical to arkworks' msm_bigint(..).into_affine()
// (reference: pcs/src/univariate_kzg/mod.rs:114-118).
//
// Packed affine point in memory: x limbs || y limbs (Montgomery), identity = all zero
// (neither curve contains (0,0) since b != 0).
#pragma once
#include "fp.hpp"

namespace jf {

struct Bn254G1 {
    using Fq = Bn254Fq;
    using Fr = Bn254Fr;
    static constexpr int ID = 0;
    static constexpr int SCALAR_BITS = 254;
};
struct Bls381G1 {
    using Fq = Bls381Fq;
    using Fr = Bls381Fr;
    static constexpr int ID = 1;
    static constexpr int SCALAR_BITS = 255;
};

template <class Cv>
struct Affine {
    Fp<typename Cv::Fq> x, y;
    JF_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
};

template <class Cv>
struct XYZZ {
    using F = Fp<typename Cv::Fq>;
    F X, Y, ZZ, ZZZ;
    JF_HD static XYZZ identity() {
        XYZZ r;
        r.X = F::zero();
        r.Y = F::zero();
        r.ZZ = F::zero();
        r.ZZZ = F::zero();
        return r;
    }
    JF_HD bool is_inf() const { return ZZ.is_zero(); }
};

template <class Cv>
JF_HD XYZZ<Cv> xyzz_from_affine(const Affine<Cv>& p) {
    using F = Fp<typename Cv::Fq>;
    XYZZ<Cv> r;
    if (p.is_inf()) return XYZZ<Cv>::identity();
    r.X = p.x;
    r.Y = p.y;
    r.ZZ = F::one();
    r.ZZZ = F::one();
    return r;
}

// 2 * (affine p), p != identity   (EFD mdbl-2008-s-1, a = 0)
template <class Cv>
JF_COLD XYZZ<Cv> xyzz_double_affine(Affine<Cv> p) {
    using Q = typename Cv::Fq;
    XYZZ<Cv> r;
    auto U = fp_dbl<Q>(p.y);
    auto V = fp_sqr<Q>(U);
    auto W = fp_mul<Q>(U, V);
    auto S = fp_mul<Q>(p.x, V);
    auto XX = fp_sqr<Q>(p.x);
    auto M = fp_add<Q>(fp_dbl<Q>(XX), XX);
    r.X = fp_sub<Q>(fp_sqr<Q>(M), fp_dbl<Q>(S));
    r.Y = fp_sub<Q>(fp_mul<Q>(M, fp_sub<Q>(S, r.X)), fp_mul<Q>(W, p.y));
    r.ZZ = V;
    r.ZZZ = W;
    return r;
}

// 2 * p   (EFD dbl-2008-s-1, a = 0); identity and y = 0 map to identity
template <class Cv>
JF_COLD XYZZ<Cv> xyzz_double(XYZZ<Cv> p) {
    using Q = typename Cv::Fq;
    if (p.is_inf() || p.Y.is_zero()) return XYZZ<Cv>::identity();
    XYZZ<Cv> r;
    auto U = fp_dbl<Q>(p.Y);
    auto V = fp_sqr<Q>(U);
    auto W = fp_mul<Q>(U, V);
    auto S = fp_mul<Q>(p.X, V);
    auto XX = fp_sqr<Q>(p.X);
    auto M = fp_add<Q>(fp_dbl<Q>(XX), XX);
    r.X = fp_sub<Q>(fp_sqr<Q>(M), fp_dbl<Q>(S));
    r.Y = fp_sub<Q>(fp_mul<Q>(M, fp_sub<Q>(S, r.X)), fp_mul<Q>(W, p.Y));
    r.ZZ = fp_mul<Q>(V, p.ZZ);
    r.ZZZ = fp_mul<Q>(W, p.ZZZ);
    return r;
}

// acc += q (affine, q may be the identity)   (EFD madd-2008-s)
template <class Cv>
JF_HD void xyzz_add_affine(XYZZ<Cv>& acc, const Affine<Cv>& q) {
    using Q = typename Cv::Fq;
    if (q.is_inf()) return;
    if (acc.is_inf()) {
        acc = xyzz_from_affine<Cv>(q);
        return;
    }
    auto U2 = fp_mul<Q>(q.x, acc.ZZ);
    auto S2 = fp_mul<Q>(q.y, acc.ZZZ);
    auto P = fp_sub<Q>(U2, acc.X);
    auto R = fp_sub<Q>(S2, acc.Y);
    if (P.is_zero()) {
        // rare (P + P or P - P).  The out-of-line call gets a COPY: passing q itself would make
        // its address escape and pin it (64 B of scratch stores per loop iteration, measured as
        // 12.9 GB of HBM writes per 2^24 MSM).
        if (R.is_zero()) {
            Affine<Cv> qc = q;
            XYZZ<Cv> d = xyzz_double_affine<Cv>(qc);
            acc = d;
        } else acc = XYZZ<Cv>::identity();
        return;
    }
    auto PP = fp_sqr<Q>(P);
    auto PPP = fp_mul<Q>(P, PP);
    auto Qv = fp_mul<Q>(acc.X, PP);
    auto X3 = fp_sub<Q>(fp_sub<Q>(fp_sqr<Q>(R), PPP), fp_dbl<Q>(Qv));
    auto Y3 = fp_sub<Q>(fp_mul<Q>(R, fp_sub<Q>(Qv, X3)), fp_mul<Q>(acc.Y, PPP));
    acc.X = X3;
    acc.Y = Y3;
    acc.ZZ = fp_mul<Q>(acc.ZZ, PP);
    acc.ZZZ = fp_mul<Q>(acc.ZZZ, PPP);
}

// acc += q   (EFD add-2008-s).  Body shared by the out-of-line and the inline entry points.
template <class Cv>
JF_HD void xyzz_add_body(XYZZ<Cv>& acc, const XYZZ<Cv>& q) {
    using Q = typename Cv::Fq;
    if (q.is_inf()) return;
    if (acc.is_inf()) {
        acc = q;
        return;
    }
    auto U1 = fp_mul<Q>(acc.X, q.ZZ);
    auto U2 = fp_mul<Q>(q.X, acc.ZZ);
    auto S1 = fp_mul<Q>(acc.Y, q.ZZZ);
    auto S2 = fp_mul<Q>(q.Y, acc.ZZZ);
    auto P = fp_sub<Q>(U2, U1);
    auto R = fp_sub<Q>(S2, S1);
    if (P.is_zero()) {
        if (R.is_zero()) {
            XYZZ<Cv> ac = acc;
            XYZZ<Cv> d = xyzz_double<Cv>(ac);
            acc = d;
        } else acc = XYZZ<Cv>::identity();
        return;
    }
    auto PP = fp_sqr<Q>(P);
    auto PPP = fp_mul<Q>(P, PP);
    auto Qv = fp_mul<Q>(U1, PP);
    auto X3 = fp_sub<Q>(fp_sub<Q>(fp_sqr<Q>(R), PPP), fp_dbl<Q>(Qv));
    auto Y3 = fp_sub<Q>(fp_mul<Q>(R, fp_sub<Q>(Qv, X3)), fp_mul<Q>(S1, PPP));
    acc.X = X3;
    acc.Y = Y3;
    acc.ZZ = fp_mul<Q>(fp_mul<Q>(acc.ZZ, q.ZZ), PP);
    acc.ZZZ = fp_mul<Q>(fp_mul<Q>(acc.ZZZ, q.ZZZ), PPP);
}
// out of line: cold call sites (keeps code size and compile time down)
template <class Cv>
JF_COLD void xyzz_add(XYZZ<Cv>& acc, const XYZZ<Cv>& q) {
    xyzz_add_body<Cv>(acc, q);
}
// inline: the latency-bound reduction kernels, where a call's scratch round trip is visible
template <class Cv>
JF_DEV void xyzz_add_inl(XYZZ<Cv>& acc, const XYZZ<Cv>& q) {
    xyzz_add_body<Cv>(acc, q);
}

// exact affine normalisation: one field inversion
template <class Cv>
JF_COLD Affine<Cv> xyzz_to_affine(const XYZZ<Cv>& p) {
    using Q = typename Cv::Fq;
    Affine<Cv> r;
    if (p.is_inf()) {
        r.x = Fp<Q>::zero();
        r.y = Fp<Q>::zero();
        return r;
    }
    auto t = fp_inv_safegcd<Q>(fp_mul<Q>(p.ZZ, p.ZZZ));
    r.x = fp_mul<Q>(p.X, fp_mul<Q>(t, p.ZZZ));
    r.y = fp_mul<Q>(p.Y, fp_mul<Q>(t, p.ZZ));
    return r;
}

// ---- memory helpers ----------------------------------------------------------------------
template <class Cv>
JF_DEV Affine<Cv> affine_load(const uint32_t* __restrict__ p) {
    constexpr int N = Cv::Fq::N;
    Affine<Cv> a;
    a.x = fp_load<typename Cv::Fq>(p);
    a.y = fp_load<typename Cv::Fq>(p + N);
    return a;
}
template <class Cv>
JF_DEV void affine_store(uint32_t* __restrict__ p, const Affine<Cv>& a) {
    constexpr int N = Cv::Fq::N;
    fp_store<typename Cv::Fq>(p, a.x);
    fp_store<typename Cv::Fq>(p + N, a.y);
}
template <class Cv>
JF_DEV XYZZ<Cv> xyzz_load(const uint32_t* __restrict__ p) {
    constexpr int N = Cv::Fq::N;
    XYZZ<Cv> a;
    a.X = fp_load<typename Cv::Fq>(p);
    a.Y = fp_load<typename Cv::Fq>(p + N);
    a.ZZ = fp_load<typename Cv::Fq>(p + 2 * N);
    a.ZZZ = fp_load<typename Cv::Fq>(p + 3 * N);
    return a;
}
template <class Cv>
JF_DEV void xyzz_store(uint32_t* __restrict__ p, const XYZZ<Cv>& a) {
    constexpr int N = Cv::Fq::N;
    fp_store<typename Cv::Fq>(p, a.X);
    fp_store<typename Cv::Fq>(p + N, a.Y);
    fp_store<typename Cv::Fq>(p + 2 * N, a.ZZ);
    fp_store<typename Cv::Fq>(p + 3 * N, a.ZZZ);
}

}  // namespace jf
