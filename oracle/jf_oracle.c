/* TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the jellyfish PLONK proving
 * hot path: KZG commit / open (Pippenger MSM) and radix-2 (coset) NTT.
 *
 * It is NOT part of the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py load the library built from this file, as the checker.
 *
 * The arithmetic of this path lives in arkworks 0.5 (ark-ff / ark-ec / ark-poly,
 * crates.io, semver "0.5", unpinned: reference Cargo.toml:30-42) which is absent from
 * /root/reference; this file restates the published algorithms and anchors semantics
 * on the reference's call sites:
 *   commit   pcs/src/univariate_kzg/mod.rs:90-125   (msm_bigint over powers_of_g)
 *   open     pcs/src/univariate_kzg/mod.rs:144-180  (divide by X - z, commit, evaluate)
 *   SRS      pcs/src/univariate_kzg/srs.rs:175-225  ([beta^i] g)
 *   (i)FFT   plonk/src/proof_system/prover.rs:524-573,658-660 and
 *            relation/src/constraint_system.rs:1369-1622 (Radix2EvaluationDomain,
 *            natural order in/out, coset offset = Fr::GENERATOR)
 *   limbs    pcs/src/univariate_kzg/mod.rs:970-998  (Montgomery u64 limbs, identity)
 * Pinning: no literal vectors exist in the reference for this path; this
 * restatement is pinned by tests/golden/ (big-integer Python, oracle/pyref.py) and by
 * the algebraic identities the reference tests assert.  Byte-level parity with
 * arkworks itself: "parity unpinned" (no Rust toolchain in the build container).
 *
 * Boundary data format (same as include/jfhip.h): Montgomery little-endian u64 limbs;
 * packed affine point = x || y, identity = all-zero.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define NL 4
#define PFX(n) f4_##n
#include "fp_tmpl.h"
#undef NL
#undef PFX
#define NL 6
#define PFX(n) f6_##n
#include "fp_tmpl.h"
#undef NL
#undef PFX

#define QL 4
#define FQ(n) f4_##n
#define EC(n) ec4_##n
#include "ec_tmpl.h"
#undef QL
#undef FQ
#undef EC
#define QL 6
#define FQ(n) f6_##n
#define EC(n) ec6_##n
#include "ec_tmpl.h"
#undef QL
#undef FQ
#undef EC

/* ---- constants (public curve parameters) ------------------------------------- */
static const uint64_t BN254_FR_P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL,
                                       0x30644e72e131a029ULL};
static const uint64_t BN254_FQ_P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL,
                                       0x30644e72e131a029ULL};
static const uint64_t BLS_FR_P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                                     0x73eda753299d7d48ULL};
static const uint64_t BLS_FQ_P[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                     0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
/* Fr::TWO_ADIC_ROOT_OF_UNITY, raw integers */
static const uint64_t BN254_ROOT[4] = {0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL,
                                       0x2a3c09f0a58a7e85ULL};
static const uint64_t BLS_ROOT[4] = {0x3829971f439f0d2bULL, 0xb63683508c2280b9ULL, 0xd09b681922c813b4ULL,
                                     0x16a2a19edfe81f20ULL};

static f4_ctx F4[3]; /* 0 bn254_fr, 1 bn254_fq, 2 bls_fr */
static f6_ctx F6;    /* bls_fq (field id 3) */
static int g_init = 0;

static uint64_t neg_inv64(uint64_t p0) {
    uint64_t x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - p0 * x;
    return (uint64_t)0 - x;
}
static void f4_setup(f4_ctx *f, const uint64_t *p) {
    memcpy(f->p, p, 32);
    f->inv = neg_inv64(p[0]);
    /* R mod p by 256 modular doublings of 1; R^2 by 256 more */
    uint64_t t[4] = {1, 0, 0, 0};
    for (int i = 0; i < 512; i++) {
        uint64_t c = f4_add_raw(t, t, t);
        if (c || f4_geq(t, f->p)) f4_sub_raw(t, t, f->p);
        if (i == 255) memcpy(f->r1, t, 32);
    }
    memcpy(f->r2, t, 32);
}
static void f6_setup(f6_ctx *f, const uint64_t *p) {
    memcpy(f->p, p, 48);
    f->inv = neg_inv64(p[0]);
    uint64_t t[6] = {1, 0, 0, 0, 0, 0};
    for (int i = 0; i < 768; i++) {
        uint64_t c = f6_add_raw(t, t, t);
        if (c || f6_geq(t, f->p)) f6_sub_raw(t, t, f->p);
        if (i == 383) memcpy(f->r1, t, 48);
    }
    memcpy(f->r2, t, 48);
}
static void ensure_init(void) {
    if (g_init) return;
#pragma omp critical(jfo_init)
    {
        if (!g_init) {
            f4_setup(&F4[0], BN254_FR_P);
            f4_setup(&F4[1], BN254_FQ_P);
            f4_setup(&F4[2], BLS_FR_P);
            f6_setup(&F6, BLS_FQ_P);
            g_init = 1;
        }
    }
}
static const f4_ctx *fr_of(int curve) { return curve == 0 ? &F4[0] : &F4[2]; }

/* ---- field vector ops ---------------------------------------------------------- */
/* field_id: 0 bn254_fr, 1 bn254_fq, 2 bls12_381_fr, 3 bls12_381_fq */
int jfo_field_limbs(int field_id) { return field_id == 3 ? 6 : 4; }

/* op: 0 add, 1 sub, 2 mul */
int jfo_fp_binop(int field_id, int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) {
    ensure_init();
    if (field_id < 0 || field_id > 3 || op < 0 || op > 2) return -1;
    for (size_t i = 0; i < n; i++) {
        if (field_id == 3) {
            const uint64_t *x = a + 6 * i, *y = b + 6 * i;
            uint64_t *o = out + 6 * i;
            if (op == 0) f6_add(&F6, o, x, y);
            else if (op == 1) f6_sub(&F6, o, x, y);
            else f6_mul(&F6, o, x, y);
        } else {
            const f4_ctx *f = &F4[field_id];
            const uint64_t *x = a + 4 * i, *y = b + 4 * i;
            uint64_t *o = out + 4 * i;
            if (op == 0) f4_add(f, o, x, y);
            else if (op == 1) f4_sub(f, o, x, y);
            else f4_mul(f, o, x, y);
        }
    }
    return 0;
}
/* op: 0 neg, 1 inv (inv(0)=0), 2 sqr, 3 from_mont, 4 to_mont */
int jfo_fp_unop(int field_id, int op, const uint64_t *a, uint64_t *out, size_t n) {
    ensure_init();
    if (field_id < 0 || field_id > 3 || op < 0 || op > 4) return -1;
    for (size_t i = 0; i < n; i++) {
        if (field_id == 3) {
            const uint64_t *x = a + 6 * i;
            uint64_t *o = out + 6 * i;
            switch (op) {
            case 0: f6_neg(&F6, o, x); break;
            case 1: f6_inv(&F6, o, x); break;
            case 2: f6_sqr(&F6, o, x); break;
            case 3: f6_from_mont(&F6, o, x); break;
            default: f6_to_mont(&F6, o, x); break;
            }
        } else {
            const f4_ctx *f = &F4[field_id];
            const uint64_t *x = a + 4 * i;
            uint64_t *o = out + 4 * i;
            switch (op) {
            case 0: f4_neg(f, o, x); break;
            case 1: f4_inv(f, o, x); break;
            case 2: f4_sqr(f, o, x); break;
            case 3: f4_from_mont(f, o, x); break;
            default: f4_to_mont(f, o, x); break;
            }
        }
    }
    return 0;
}

/* ---- group ops -------------------------------------------------------------------- */
int jfo_ec_add(int curve, const uint64_t *P, const uint64_t *Q, uint64_t *out) {
    ensure_init();
    if (curve == 0) {
        ec4_jac a, r;
        ec4_jac_from_aff(&F4[1], &a, P);
        ec4_jac_add_aff(&F4[1], &r, &a, Q, 0);
        ec4_jac_to_aff(&F4[1], out, &r);
    } else if (curve == 1) {
        ec6_jac a, r;
        ec6_jac_from_aff(&F6, &a, P);
        ec6_jac_add_aff(&F6, &r, &a, Q, 0);
        ec6_jac_to_aff(&F6, out, &r);
    } else return -1;
    return 0;
}
/* out = k * P, k a Montgomery Fr element */
int jfo_ec_mul(int curve, const uint64_t *P, const uint64_t *k_mont, uint64_t *out) {
    ensure_init();
    if (curve != 0 && curve != 1) return -1;
    uint64_t k[4];
    f4_from_mont(fr_of(curve), k, k_mont);
    if (curve == 0) {
        ec4_jac a, r;
        ec4_jac_from_aff(&F4[1], &a, P);
        ec4_jac_mul(&F4[1], &r, &a, k, 4);
        ec4_jac_to_aff(&F4[1], out, &r);
    } else {
        ec6_jac a, r;
        ec6_jac_from_aff(&F6, &a, P);
        ec6_jac_mul(&F6, &r, &a, k, 4);
        ec6_jac_to_aff(&F6, out, &r);
    }
    return 0;
}
/* 1 if the packed affine point satisfies y^2 = x^3 + b (identity counts as on-curve) */
int jfo_ec_on_curve(int curve, const uint64_t *P) {
    ensure_init();
    if (curve == 0) {
        if (ec4_aff_is_inf(P)) return 1;
        uint64_t l[4], r[4], b[4];
        f4_sqr(&F4[1], l, P + 4);
        f4_sqr(&F4[1], r, P);
        f4_mul(&F4[1], r, r, P);
        f4_from_u64(&F4[1], b, 3);
        f4_add(&F4[1], r, r, b);
        return f4_eq(l, r);
    } else {
        if (ec6_aff_is_inf(P)) return 1;
        uint64_t l[6], r[6], b[6];
        f6_sqr(&F6, l, P + 6);
        f6_sqr(&F6, r, P);
        f6_mul(&F6, r, r, P);
        f6_from_u64(&F6, b, 4);
        f6_add(&F6, r, r, b);
        return f6_eq(l, r);
    }
}

/* SRS powers [beta^i] g, i < n  (srs.rs:175-225).  Fixed-base 8-bit windows. */
#define SRS_GEN_IMPL(QL_, FQT, FQCTX, ECP)                                                              \
    {                                                                                                     \
        const int WB = 8, NW = 32;                                                                        \
        ECP##jac *tab = (ECP##jac *)malloc(sizeof(ECP##jac) * NW * 256);                                  \
        ECP##jac base;                                                                                    \
        ECP##jac_from_aff(FQCTX, &base, g_xy);                                                            \
        for (int w = 0; w < NW; w++) {                                                                    \
            ECP##jac_set_inf(FQCTX, &tab[w * 256]);                                                       \
            for (int d = 1; d < 256; d++) ECP##jac_add(FQCTX, &tab[w * 256 + d], &tab[w * 256 + d - 1], &base); \
            for (int k = 0; k < WB; k++) ECP##jac_double(FQCTX, &base, &base);                            \
        }                                                                                                 \
        uint64_t *taff = (uint64_t *)malloc(sizeof(uint64_t) * 2 * QL_ * NW * 256);                       \
        ECP##batch_to_aff(FQCTX, taff, tab, (size_t)NW * 256);                                            \
        free(tab);                                                                                        \
        const size_t CH = 4096;                                                                           \
        size_t nch = (n + CH - 1) / CH;                                                                   \
        _Pragma("omp parallel for schedule(dynamic,1) num_threads(threads)")                              \
        for (size_t ch = 0; ch < nch; ch++) {                                                             \
            size_t lo = ch * CH, hi = lo + CH < n ? lo + CH : n;                                          \
            ECP##jac *pts = (ECP##jac *)malloc(sizeof(ECP##jac) * (hi - lo));                             \
            uint64_t s[4], e[1] = {lo}, sr[4];                                                            \
            f4_pow(fr, s, beta_mont, e, 1);                                                               \
            for (size_t i = lo; i < hi; i++) {                                                            \
                f4_from_mont(fr, sr, s);                                                                  \
                ECP##jac acc;                                                                             \
                ECP##jac_set_inf(FQCTX, &acc);                                                            \
                for (int w = 0; w < NW; w++) {                                                            \
                    unsigned d = (unsigned)(sr[w >> 3] >> ((w & 7) * 8)) & 0xff;                          \
                    if (d) ECP##jac_add_aff(FQCTX, &acc, &acc, taff + (size_t)(w * 256 + d) * 2 * QL_, 0); \
                }                                                                                         \
                pts[i - lo] = acc;                                                                        \
                f4_mul(fr, s, s, beta_mont);                                                              \
            }                                                                                             \
            ECP##batch_to_aff(FQCTX, out + lo * 2 * QL_, pts, hi - lo);                                   \
            free(pts);                                                                                    \
        }                                                                                                 \
        free(taff);                                                                                       \
    }

int jfo_srs_gen(int curve, const uint64_t *beta_mont, const uint64_t *g_xy, size_t n, uint64_t *out, int threads) {
    ensure_init();
    if (curve != 0 && curve != 1) return -1;
    if (threads < 1) threads = 1;
    const f4_ctx *fr = fr_of(curve);
    if (curve == 0) SRS_GEN_IMPL(4, f4_, &F4[1], ec4_)
    else SRS_GEN_IMPL(6, f6_, &F6, ec6_)
    return 0;
}

/* MSM: out = sum_i scalars[i] * bases[i]  (affine packed, identity = zeros) */
int jfo_msm(int curve, const uint64_t *bases, const uint64_t *scalars_mont, size_t n, uint64_t *out_xy, int threads) {
    ensure_init();
    if (curve != 0 && curve != 1) return -1;
    if (threads < 1) threads = 1;
    const f4_ctx *fr = fr_of(curve);
    uint64_t *raw = (uint64_t *)malloc(32 * (n ? n : 1));
#pragma omp parallel for num_threads(threads)
    for (size_t i = 0; i < n; i++) f4_from_mont(fr, raw + 4 * i, scalars_mont + 4 * i);
    int bits = curve == 0 ? 254 : 255;
    if (curve == 0) {
        ec4_jac r;
        ec4_msm_pippenger(&F4[1], &r, bases, raw, n, bits, threads);
        ec4_jac_to_aff(&F4[1], out_xy, &r);
    } else {
        ec6_jac r;
        ec6_msm_pippenger(&F6, &r, bases, raw, n, bits, threads);
        ec6_jac_to_aff(&F6, out_xy, &r);
    }
    free(raw);
    return 0;
}

/* definition-level MSM (double-and-add per term); cross-check for the Pippenger above */
int jfo_msm_naive(int curve, const uint64_t *bases, const uint64_t *scalars_mont, size_t n, uint64_t *out_xy) {
    ensure_init();
    if (curve != 0 && curve != 1) return -1;
    const f4_ctx *fr = fr_of(curve);
    if (curve == 0) {
        ec4_jac acc, b, t;
        ec4_jac_set_inf(&F4[1], &acc);
        for (size_t i = 0; i < n; i++) {
            uint64_t k[4];
            f4_from_mont(fr, k, scalars_mont + 4 * i);
            ec4_jac_from_aff(&F4[1], &b, bases + 8 * i);
            ec4_jac_mul(&F4[1], &t, &b, k, 4);
            ec4_jac_add(&F4[1], &acc, &acc, &t);
        }
        ec4_jac_to_aff(&F4[1], out_xy, &acc);
    } else {
        ec6_jac acc, b, t;
        ec6_jac_set_inf(&F6, &acc);
        for (size_t i = 0; i < n; i++) {
            uint64_t k[4];
            f4_from_mont(fr, k, scalars_mont + 4 * i);
            ec6_jac_from_aff(&F6, &b, bases + 12 * i);
            ec6_jac_mul(&F6, &t, &b, k, 4);
            ec6_jac_add(&F6, &acc, &acc, &t);
        }
        ec6_jac_to_aff(&F6, out_xy, &acc);
    }
    return 0;
}

/* ---- polynomials over Fr ------------------------------------------------------------ */
/* Horner evaluation (mod.rs:174) */
int jfo_poly_eval(int curve, const uint64_t *coeffs, size_t n, const uint64_t *z, uint64_t *out) {
    ensure_init();
    if (curve != 0 && curve != 1) return -1;
    const f4_ctx *fr = fr_of(curve);
    uint64_t acc[4] = {0, 0, 0, 0};
    for (size_t i = n; i-- > 0;) {
        f4_mul(fr, acc, acc, z);
        f4_add(fr, acc, acc, coeffs + 4 * i);
    }
    memcpy(out, acc, 32);
    return 0;
}
/* witness polynomial (f - f(z)) / (X - z) (mod.rs:153-158): q has n-1 coefficients; eval = f(z) */
int jfo_open_witness(int curve, const uint64_t *coeffs, size_t n, const uint64_t *z, uint64_t *q_out,
                     uint64_t *eval_out) {
    ensure_init();
    if (curve != 0 && curve != 1) return -1;
    const f4_ctx *fr = fr_of(curve);
    uint64_t carry[4] = {0, 0, 0, 0};
    if (n == 0) { memset(eval_out, 0, 32); return 0; }
    for (size_t i = n - 1; i >= 1; i--) {
        f4_mul(fr, carry, carry, z);
        f4_add(fr, carry, carry, coeffs + 4 * i);
        memcpy(q_out + 4 * (i - 1), carry, 32);
    }
    f4_mul(fr, carry, carry, z);
    f4_add(fr, carry, carry, coeffs);
    memcpy(eval_out, carry, 32);
    return 0;
}

/* ---- radix-2 (coset) NTT ---------------------------------------------------------- */
static void domain_gen(int curve, int log_n, uint64_t *w_mont) {
    const f4_ctx *fr = fr_of(curve);
    int s = curve == 0 ? 28 : 32;
    uint64_t t[4];
    f4_to_mont(fr, t, curve == 0 ? BN254_ROOT : BLS_ROOT);
    for (int i = 0; i < s - log_n; i++) f4_sqr(fr, t, t);
    memcpy(w_mont, t, 32);
}
int jfo_domain_root(int curve, int log_n, uint64_t *w_mont) {
    ensure_init();
    if ((curve != 0 && curve != 1) || log_n < 0 || log_n > (curve == 0 ? 28 : 32)) return -1;
    domain_gen(curve, log_n, w_mont);
    return 0;
}

static size_t bitrev(size_t x, int bits) {
    size_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

/* In-place, natural order in and out.  data: batch polys of 2^log_n Montgomery elements.
 * inverse = 0: out[i] = sum_j in[j] (h w^i)^j ; inverse = 1: the inverse map (scale by
 * n^{-1} h^{-j}).  coset_mont = NULL means h = 1. */
int jfo_ntt(int curve, uint64_t *data, int log_n, size_t batch, int inverse, const uint64_t *coset_mont,
            int threads) {
    ensure_init();
    if ((curve != 0 && curve != 1) || log_n < 0 || log_n > (curve == 0 ? 28 : 32)) return -1;
    if (threads < 1) threads = 1;
    const f4_ctx *fr = fr_of(curve);
    size_t n = (size_t)1 << log_n;
    uint64_t w[4];
    domain_gen(curve, log_n, w);
    if (inverse) f4_inv(fr, w, w);
    /* twiddles w^k, k < n/2 */
    size_t half = n / 2 ? n / 2 : 1;
    uint64_t *tw = (uint64_t *)malloc(32 * half);
    {
        /* chunked so it parallelises */
        size_t CH = 4096;
        size_t nch = (half + CH - 1) / CH;
#pragma omp parallel for num_threads(threads)
        for (size_t c = 0; c < nch; c++) {
            size_t lo = c * CH, hi = lo + CH < half ? lo + CH : half;
            uint64_t t[4], e[1] = {lo};
            f4_pow(fr, t, w, e, 1);
            for (size_t k = lo; k < hi; k++) {
                memcpy(tw + 4 * k, t, 32);
                f4_mul(fr, t, t, w);
            }
        }
    }
    uint64_t *cs = NULL; /* per-index scale factors */
    if (coset_mont || inverse) {
        cs = (uint64_t *)malloc(32 * n);
        uint64_t step[4], start[4];
        if (coset_mont) memcpy(step, coset_mont, 32); else memcpy(step, fr->r1, 32);
        memcpy(start, fr->r1, 32);
        if (inverse) {
            f4_inv(fr, step, step);
            uint64_t nn[4];
            f4_from_u64(fr, nn, (uint64_t)n);
            f4_inv(fr, start, nn);
        }
        size_t CH = 4096;
        size_t nch = (n + CH - 1) / CH;
#pragma omp parallel for num_threads(threads)
        for (size_t c = 0; c < nch; c++) {
            size_t lo = c * CH, hi = lo + CH < n ? lo + CH : n;
            uint64_t t[4], e[1] = {lo};
            f4_pow(fr, t, step, e, 1);
            f4_mul(fr, t, t, start);
            for (size_t k = lo; k < hi; k++) {
                memcpy(cs + 4 * k, t, 32);
                f4_mul(fr, t, t, step);
            }
        }
    }
    int inner_threads = 1, outer_threads = threads;
    if (batch < (size_t)threads) { outer_threads = (int)batch; inner_threads = threads / (int)batch; }
    if (inner_threads < 1) inner_threads = 1;
    omp_set_max_active_levels(2);
#pragma omp parallel for num_threads(outer_threads) schedule(dynamic, 1)
    for (size_t b = 0; b < batch; b++) {
        uint64_t *a = data + 4 * n * b;
        if (!inverse && cs) {
#pragma omp parallel for num_threads(inner_threads)
            for (size_t i = 0; i < n; i++) f4_mul(fr, a + 4 * i, a + 4 * i, cs + 4 * i);
        }
        for (size_t i = 0; i < n; i++) {
            size_t j = bitrev(i, log_n);
            if (i < j) {
                uint64_t t[4];
                memcpy(t, a + 4 * i, 32);
                memcpy(a + 4 * i, a + 4 * j, 32);
                memcpy(a + 4 * j, t, 32);
            }
        }
        for (int s = 1; s <= log_n; s++) {
            size_t m = (size_t)1 << s, mh = m >> 1, stride = n / m;
#pragma omp parallel for num_threads(inner_threads)
            for (size_t q = 0; q < n / 2; q++) {
                size_t grp = q / mh, j = q % mh;
                uint64_t *u = a + 4 * (grp * m + j), *v = u + 4 * mh;
                uint64_t t[4];
                f4_mul(fr, t, v, tw + 4 * (j * stride));
                f4_sub(fr, v, u, t);
                f4_add(fr, u, u, t);
            }
        }
        if (inverse) {
#pragma omp parallel for num_threads(inner_threads)
            for (size_t i = 0; i < n; i++) f4_mul(fr, a + 4 * i, a + 4 * i, cs + 4 * i);
        }
    }
    free(tw);
    if (cs) free(cs);
    return 0;
}

int jfo_threads(void) { return omp_get_max_threads(); }

/* ---- TurboPlonk quotient evaluations -----------------------------------
 * Restates plonk/src/proof_system/prover.rs:587-655 with :665-699 (gate part) and :709-752
 * (copy-constraint part) for one instance without Plookup:
 *   t_i = (t_circ + t_perm_1) * z_h_inv[i % ratio] + t_perm_2,   out[i] += alpha_base * t_i
 * All arrays are coset evaluations over the size-m quotient domain (coset offset = Fr::GENERATOR),
 * laid out [poly][m]; 13 selectors in the reference's order q_lc[4], q_mul[2], q_hash[4], q_o, q_c,
 * q_ecc; nw wire types (5); k[nw] coset representatives; every element Montgomery. */
static int quot_core(int curve, int log_n, int log_m, int nw, const uint64_t *sel, const uint64_t *sig,
                     const uint64_t *wires, const uint64_t *zperm, const uint64_t *pi, const uint64_t *lookup,
                     const uint64_t *k, const uint64_t *alpha, const uint64_t *beta, const uint64_t *gamma,
                     const uint64_t *tau, const uint64_t *alpha_base, uint64_t *out, int threads) {
    ensure_init();
    if ((curve != 0 && curve != 1) || (nw != 5 && nw != 6) || log_m < log_n) return -1;
    if ((nw == 6) != (lookup != NULL)) return -1;
    if (threads < 1) threads = 1;
    const f4_ctx *fr = fr_of(curve);
    const size_t n = (size_t)1 << log_n, m = (size_t)1 << log_m, ratio = m / n;
    uint64_t wm[4], g[4], one[4], nn[4], alpha2[4];
    domain_gen(curve, log_m, wm);
    f4_from_u64(fr, g, curve == 0 ? 5 : 7);
    memcpy(one, fr->r1, 32);
    f4_from_u64(fr, nn, (uint64_t)n);
    f4_sqr(fr, alpha2, alpha);
    uint64_t wn_inv[4];
    domain_gen(curve, log_n, wn_inv);
    f4_inv(fr, wn_inv, wn_inv);
    /* z_h_inv[i] = 1 / ((g w^i)^n - 1), i < ratio   (prover.rs:509-516) */
    uint64_t *zh = (uint64_t *)malloc(32 * ratio);
    for (size_t i = 0; i < ratio; i++) {
        uint64_t x[4], e[1] = {i};
        f4_pow(fr, x, wm, e, 1);
        f4_mul(fr, x, x, g);
        e[0] = n;
        f4_pow(fr, x, x, e, 1);
        f4_sub(fr, x, x, one);
        f4_inv(fr, zh + 4 * i, x);
    }
#pragma omp parallel for num_threads(threads) schedule(static)
    for (size_t i = 0; i < m; i++) {
        uint64_t x[4], e[1] = {i}, t[4], u[4], acc[4], w[6][4];
        f4_pow(fr, x, wm, e, 1);
        f4_mul(fr, x, x, g); /* eval_point */
        for (int j = 0; j < nw; j++) memcpy(w[j], wires + 4 * (j * m + i), 32);
#define SEL(j) (sel + 4 * ((size_t)(j) * m + i))
        /* gate part :665-699 */
        f4_add(fr, acc, SEL(11), pi + 4 * i); /* q_c + pi */
        for (int j = 0; j < 4; j++) {
            f4_mul(fr, t, SEL(j), w[j]);
            f4_add(fr, acc, acc, t);
        }
        f4_mul(fr, t, SEL(4), w[0]);
        f4_mul(fr, t, t, w[1]);
        f4_add(fr, acc, acc, t);
        f4_mul(fr, t, SEL(5), w[2]);
        f4_mul(fr, t, t, w[3]);
        f4_add(fr, acc, acc, t);
        f4_mul(fr, t, SEL(12), w[0]);
        for (int j = 1; j < 5; j++) f4_mul(fr, t, t, w[j]);
        f4_add(fr, acc, acc, t);
        for (int j = 0; j < 4; j++) {
            f4_sqr(fr, u, w[j]);
            f4_sqr(fr, u, u);
            f4_mul(fr, u, u, w[j]);
            f4_mul(fr, t, SEL(6 + j), u);
            f4_add(fr, acc, acc, t);
        }
        f4_mul(fr, t, SEL(10), w[4]);
        f4_sub(fr, acc, acc, t);
#undef SEL
        /* copy-constraint part :709-752 */
        const uint64_t *zx = zperm + 4 * i, *zxw = zperm + 4 * ((i + ratio) % m);
        uint64_t r1[4], r1b[4];
        memcpy(r1, zx, 32);
        for (int j = 0; j < nw; j++) {
            f4_mul(fr, t, k + 4 * j, x);
            f4_mul(fr, t, t, beta);
            f4_add(fr, t, t, w[j]);
            f4_add(fr, t, t, gamma);
            f4_mul(fr, r1, r1, t);
        }
        f4_mul(fr, r1, r1, alpha);
        memcpy(r1b, zxw, 32);
        for (int j = 0; j < nw; j++) {
            f4_mul(fr, t, sig + 4 * (j * m + i), beta);
            f4_add(fr, t, t, w[j]);
            f4_add(fr, t, t, gamma);
            f4_mul(fr, r1b, r1b, t);
        }
        f4_mul(fr, r1b, r1b, alpha);
        f4_sub(fr, r1, r1, r1b);
        uint64_t r2[4], den[4];
        f4_sub(fr, r2, zx, one);
        f4_mul(fr, r2, r2, alpha2);
        f4_sub(fr, den, x, one);
        f4_mul(fr, den, den, nn);
        f4_inv(fr, den, den);
        f4_mul(fr, r2, r2, den);
        f4_add(fr, acc, acc, r1);
        if (lookup) {
            /* Plookup part :766-882; lookup = [range_table, key_table, table_dom_sep, q_dom_sep, h1, h2, prod_lookup] */
            const size_t nx = (i + ratio) % m;
#define LK(p, idx) (lookup + 4 * ((size_t)(p) * m + (idx)))
            const uint64_t *ql = sel + 4 * ((size_t)13 * m + i), *qlw = sel + 4 * ((size_t)13 * m + nx);
            uint64_t wn3[4], wn4[4], mtx[4], mtxw[4], mlx[4], l1d[4], lnd[4], ap[4], b1[4], gb[4], s1[4], s2[4];
            memcpy(wn3, wires + 4 * ((size_t)3 * m + nx), 32);
            memcpy(wn4, wires + 4 * ((size_t)4 * m + nx), 32);
            /* eval_merged_table (structs.rs:899-914) at x and at x * w */
            f4_mul(fr, t, tau, w[4]);
            f4_add(fr, t, t, w[3]);
            f4_mul(fr, t, t, tau);
            f4_add(fr, t, t, LK(1, i));
            f4_mul(fr, t, t, tau);
            f4_add(fr, t, t, LK(2, i));
            f4_mul(fr, t, t, tau);
            f4_mul(fr, t, t, ql);
            f4_add(fr, mtx, t, LK(0, i));
            f4_mul(fr, t, tau, wn4);
            f4_add(fr, t, t, wn3);
            f4_mul(fr, t, t, tau);
            f4_add(fr, t, t, LK(1, nx));
            f4_mul(fr, t, t, tau);
            f4_add(fr, t, t, LK(2, nx));
            f4_mul(fr, t, t, tau);
            f4_mul(fr, t, t, qlw);
            f4_add(fr, mtxw, t, LK(0, nx));
            /* eval_merged_lookup_witness (structs.rs:916-931) */
            f4_mul(fr, t, tau, w[2]);
            f4_add(fr, t, t, w[1]);
            f4_mul(fr, t, t, tau);
            f4_add(fr, t, t, w[0]);
            f4_mul(fr, t, t, tau);
            f4_add(fr, t, t, LK(3, i));
            f4_mul(fr, t, t, tau);
            f4_mul(fr, t, t, ql);
            f4_add(fr, mlx, t, w[5]);
            /* L1(x)/Z_H(x) = 1/(n (x - 1)),  Ln(x)/Z_H(x) = w^-1 / (n (x - w^-1))   (lagrange.rs:89-115) */
            f4_sub(fr, l1d, x, one);
            f4_mul(fr, l1d, l1d, nn);
            f4_inv(fr, l1d, l1d);
            f4_sub(fr, lnd, x, wn_inv);
            f4_mul(fr, lnd, lnd, nn);
            f4_inv(fr, lnd, lnd);
            f4_mul(fr, lnd, lnd, wn_inv);
            uint64_t res2[4], p1[4];
            f4_mul(fr, ap, alpha2, alpha); /* alpha^3 */
            f4_sub(fr, t, LK(4, i), LK(5, nx));
            f4_mul(fr, t, t, lnd);
            f4_mul(fr, res2, t, ap);
            f4_mul(fr, ap, ap, alpha);
            f4_sub(fr, p1, LK(6, i), one);
            f4_mul(fr, t, p1, l1d);
            f4_mul(fr, t, t, ap);
            f4_add(fr, res2, res2, t);
            f4_mul(fr, ap, ap, alpha);
            f4_mul(fr, t, p1, lnd);
            f4_mul(fr, t, t, ap);
            f4_add(fr, res2, res2, t);
            f4_mul(fr, ap, ap, alpha); /* alpha^6 */
            f4_add(fr, b1, one, beta);
            f4_mul(fr, gb, b1, gamma);
            /* p(x)(1+beta)(gamma + ml)(gb + mt_x + beta mt_xw) */
            f4_mul(fr, s1, LK(6, i), b1);
            f4_add(fr, t, gamma, mlx);
            f4_mul(fr, s1, s1, t);
            f4_mul(fr, t, beta, mtxw);
            f4_add(fr, t, t, mtx);
            f4_add(fr, t, t, gb);
            f4_mul(fr, s1, s1, t);
            /* p(xw)(gb + h1_x + beta h1_xw)(gb + h2_x + beta h2_xw) */
            f4_mul(fr, t, beta, LK(4, nx));
            f4_add(fr, t, t, LK(4, i));
            f4_add(fr, t, t, gb);
            f4_mul(fr, s2, LK(6, nx), t);
            f4_mul(fr, t, beta, LK(5, nx));
            f4_add(fr, t, t, LK(5, i));
            f4_add(fr, t, t, gb);
            f4_mul(fr, s2, s2, t);
            f4_sub(fr, s1, s1, s2);
            f4_sub(fr, t, x, wn_inv);
            f4_mul(fr, s1, s1, t);
            f4_mul(fr, s1, s1, ap);
            f4_add(fr, acc, acc, s1);
            f4_add(fr, r2, r2, res2);
#undef LK
        }
        /* combine :636-641, :644-655 */
        f4_mul(fr, acc, acc, zh + 4 * (i % ratio));
        f4_add(fr, acc, acc, r2);
        f4_mul(fr, acc, acc, alpha_base);
        f4_add(fr, out + 4 * i, out + 4 * i, acc);
    }
    free(zh);
    return 0;
}

int jfo_plonk_quotient_turbo(int curve, int log_n, int log_m, int nw, const uint64_t *sel, const uint64_t *sig,
                             const uint64_t *wires, const uint64_t *zperm, const uint64_t *pi, const uint64_t *k,
                             const uint64_t *alpha, const uint64_t *beta, const uint64_t *gamma,
                             const uint64_t *alpha_base, uint64_t *out, int threads) {
    if (nw != 5) return -1;
    return quot_core(curve, log_n, log_m, 5, sel, sig, wires, zperm, pi, NULL, k, alpha, beta, gamma, NULL, alpha_base,
                     out, threads);
}

/* UltraPlonk: 6 wire types, 14 selectors (q_lookup last), lookup = 7 polys [range_table, key_table,
 * table_dom_sep, q_dom_sep, h1, h2, prod_lookup]; adds prover.rs:766-882. */
int jfo_plonk_quotient_ultra(int curve, int log_n, int log_m, const uint64_t *sel, const uint64_t *sig,
                             const uint64_t *wires, const uint64_t *zperm, const uint64_t *pi, const uint64_t *lookup,
                             const uint64_t *k, const uint64_t *alpha, const uint64_t *beta, const uint64_t *gamma,
                             const uint64_t *tau, const uint64_t *alpha_base, uint64_t *out, int threads) {
    if (!lookup || !tau) return -1;
    return quot_core(curve, log_n, log_m, 6, sel, sig, wires, zperm, pi, lookup, k, alpha, beta, gamma, tau, alpha_base,
                     out, threads);
}

/* ---- grand products -------------------------------------------------------
 * Serial restatements, one division per row as the reference:
 * compute_prod_permutation_polynomial relation/src/constraint_system.rs:1394-1420 (before its ifft),
 * arrays [wire type][n]. */
int jfo_plonk_perm_product(int curve, int log_n, int nw, const uint64_t *wires, const uint64_t *idp,
                           const uint64_t *sig, const uint64_t *beta, const uint64_t *gamma, uint64_t *out) {
    ensure_init();
    if ((curve != 0 && curve != 1) || nw < 1) return -1;
    const f4_ctx *fr = fr_of(curve);
    const size_t n = (size_t)1 << log_n;
    memcpy(out, fr->r1, 32);
    for (size_t j = 0; j + 1 < n; j++) {
        uint64_t a[4], b[4], t[4], u[4];
        memcpy(a, fr->r1, 32);
        memcpy(b, fr->r1, 32);
        for (int i = 0; i < nw; i++) {
            const size_t e = 4 * ((size_t)i * n + j);
            f4_add(fr, t, wires + e, gamma);
            f4_mul(fr, u, beta, idp + e);
            f4_add(fr, u, u, t);
            f4_mul(fr, a, a, u);
            f4_mul(fr, u, beta, sig + e);
            f4_add(fr, u, u, t);
            f4_mul(fr, b, b, u);
        }
        f4_inv(fr, b, b);
        f4_mul(fr, a, a, b);
        f4_mul(fr, out + 4 * (j + 1), out + 4 * j, a);
    }
    return 0;
}

/* compute_lookup_prod_polynomial relation/src/constraint_system.rs:1516-1575 (before its ifft):
 * f[n] merged lookup witness, t[n] merged table, s[2n-1] sorted vector. */
int jfo_plonk_lookup_product(int curve, int log_n, const uint64_t *f, const uint64_t *t, const uint64_t *s,
                             const uint64_t *beta, const uint64_t *gamma, uint64_t *out) {
    ensure_init();
    if (curve != 0 && curve != 1) return -1;
    const f4_ctx *fr = fr_of(curve);
    const size_t n = (size_t)1 << log_n;
    uint64_t b1[4], gb[4];
    f4_add(fr, b1, fr->r1, beta);
    f4_mul(fr, gb, gamma, b1);
    memcpy(out, fr->r1, 32);
    for (size_t j = 0; j + 2 < n; j++) {
        uint64_t a[4], b[4], u[4];
        f4_add(fr, a, gamma, f + 4 * j);
        f4_mul(fr, a, a, b1);
        f4_mul(fr, u, beta, t + 4 * (j + 1));
        f4_add(fr, u, u, t + 4 * j);
        f4_add(fr, u, u, gb);
        f4_mul(fr, a, a, u);
        f4_mul(fr, u, beta, s + 4 * (j + 1));
        f4_add(fr, u, u, s + 4 * j);
        f4_add(fr, b, u, gb);
        f4_mul(fr, u, beta, s + 4 * (n + j));
        f4_add(fr, u, u, s + 4 * (n - 1 + j));
        f4_add(fr, u, u, gb);
        f4_mul(fr, b, b, u);
        f4_inv(fr, b, b);
        f4_mul(fr, a, a, b);
        f4_mul(fr, out + 4 * (j + 1), out + 4 * j, a);
    }
    if (n > 1) memcpy(out + 4 * (n - 1), fr->r1, 32);
    return 0;
}
