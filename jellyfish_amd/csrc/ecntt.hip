// FK23 amortised opening at roots of unity for gfx950: the group-element FFT and the Toeplitz product behind
// UnivariatePCS::multi_open_rou_proofs (reference pcs/src/univariate_kzg/mod.rs:328-395; h-polynomial :572-603 /
// :606-650; group-coefficient FFT pcs/src/poly.rs:140-155; Toeplitz product pcs/src/toeplitz.rs:43-66,122-144).
//
// For f of degree d and S_k = [beta^k] g the proofs at w^j are the DFT of
//   h_i = sum_{k = 0}^{d-1-i} f_{i+1+k} S_k,  i < d        (the matrix-vector product on page 3 of FK23),
// a correlation: with r_m = f_{d-m} it is the linear convolution (S * r) read backwards, h_i = (S * r)_{d-1-i}.  The
// convolution runs as  iDFT_M( DFT_M(S) . DFT_M(r) )  with M >= 2d: two transforms over GROUP elements (butterflies
// a +- w b, w b a scalar multiplication), one scalar NTT (ntt.hip) and M scalar multiplications; a third group
// transform over the evaluation domain gives the proofs.  Group elements are unique, so the affine results equal the
// reference's (which asserts multi_open == per-point open, mod.rs:1237-1301) whatever route computes them.
//
// This is the one new kernel shape of SURVEY.md 8f rank 4; it is latency-bound (M / 2 lanes per layer, each a
// 255-bit double-and-add) and written for exactness, not yet for speed.
#include "common.hpp"
#include "curve.hpp"
#include "fp29.hpp"
#include "g29.hpp"

namespace jf {

constexpr int EC_BLOCK = 64;

// tw[j] = w_M^(+-j) as a canonical integer, j < M / 2; (lo, hi, a): two-level table of w_M (ntt_root_tables)
template <class Cv>
__global__ __launch_bounds__(256) void k_ec_twiddles(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi,
                                                     uint32_t a, uint32_t log_M, int inverse, uint32_t* __restrict__ tw) {
    using Fr = typename Cv::Fr;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t M = 1u << log_M;
    if (j >= M / 2 && !(M == 1 && j == 0)) return;
    const uint32_t e = inverse ? ((M - j) & (M - 1)) : j;
    Fp<Fr> w = fp_load<Fr>(lo + (size_t)(e & ((1u << a) - 1u)) * Fr::N);
    if (e >> a) w = fp_mul<Fr>(w, fp_load<Fr>(hi + (size_t)(e >> a) * Fr::N));
    fp_store<Fr>(tw + (size_t)j * Fr::N, fp_from_mont<Fr>(w));
}

// out[k] = SRS point k (table row 0) as XYZZ for k < count, identity for count <= k < M
template <class Cv>
__global__ __launch_bounds__(256) void k_ec_from_srs(const uint32_t* __restrict__ table, int f29, uint32_t count, uint32_t M,
                                                     uint32_t* __restrict__ out) {
    using Q = typename Cv::Fq;
    constexpr int PT = 2 * Q::N, XZ = 4 * Q::N;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    XYZZ<Cv> p = XYZZ<Cv>::identity();
    if (k < count) {
        Affine<Cv> a = affine_load<Cv>(table + (size_t)k * PT);
        if (!a.is_inf()) {
            if (f29) {   // the table carries the 29-bit kernels' Montgomery factor (srs->f29)
                a.x = fp_scale_down_r29<Q>(a.x);
                a.y = fp_scale_down_r29<Q>(a.y);
            }
            p = xyzz_from_affine<Cv>(a);
        }
    }
    xyzz_store<Cv>(out + (size_t)k * XZ, p);
}

__device__ __forceinline__ uint32_t ec_bitrev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

template <class Cv>
__global__ __launch_bounds__(256) void k_ec_bitrev(const uint32_t* __restrict__ in, uint32_t log_M,
                                                   uint32_t* __restrict__ out) {
    constexpr int XZ = 4 * Cv::Fq::N;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (1u << log_M)) return;
    xyzz_store<Cv>(out + (size_t)ec_bitrev(k, log_M) * XZ, xyzz_load<Cv>(in + (size_t)k * XZ));
}

// decimation-in-time layer s (1-based) on bit-reversed input: (a, b) -> (a + w b, a - w b), w = w_M^(j M / 2^s)
template <class Cv>
__global__ __launch_bounds__(EC_BLOCK) void k_ec_layer(uint32_t* __restrict__ data, uint32_t log_M, uint32_t s,
                                                       const uint32_t* __restrict__ tw) {
    using Fr = typename Cv::Fr;
    using Q = typename Cv::Fq;
    constexpr int XZ = 4 * Q::N;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (1u << (log_M - 1))) return;
    const uint32_t half = 1u << (s - 1);
    const uint32_t j = t & (half - 1), grp = t >> (s - 1);
    const uint32_t i0 = (grp << s) + j, i1 = i0 + half;
    // in memory the elements stay saturated XYZZ; the butterfly runs on the 29-bit-limb group law (g29.hpp)
    P29<Cv> a = p29_from_std<Cv>(xyzz_load<Cv>(data + (size_t)i0 * XZ));
    P29<Cv> b = p29_from_std<Cv>(xyzz_load<Cv>(data + (size_t)i1 * XZ));
    P29<Cv> nb;
    if (j && !b.inf) {
        b = p29_mul_scalar<Cv>(b, fp_load<Fr>(tw + (size_t)(j << (log_M - s)) * Fr::N));
        nb = p29_neg<Cv, 2>(b);
    } else {
        nb = p29_neg<Cv, 1>(b);
    }
    P29<Cv> sum = a;
    p29_add<Cv>(sum, b);
    p29_add<Cv>(a, nb);
    xyzz_store<Cv>(data + (size_t)i0 * XZ, xyzz29_to_std<Cv>(sum.v, sum.inf));
    xyzz_store<Cv>(data + (size_t)i1 * XZ, xyzz29_to_std<Cv>(a.v, a.inf));
}

// data[k] = (scalars[k] * extra) . data[k]   (scalars, extra: Fr Montgomery)
template <class Cv>
__global__ __launch_bounds__(EC_BLOCK) void k_ec_scale(uint32_t* __restrict__ data, const uint32_t* __restrict__ scalars,
                                                       const uint32_t* __restrict__ extra, uint32_t M) {
    using Fr = typename Cv::Fr;
    constexpr int XZ = 4 * Cv::Fq::N;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    XYZZ<Cv> p = xyzz_load<Cv>(data + (size_t)k * XZ);
    if (p.is_inf()) return;
    const Fp<Fr> sc = fp_from_mont<Fr>(fp_mul<Fr>(fp_load<Fr>(scalars + (size_t)k * Fr::N), fp_load<Fr>(extra)));
    const P29<Cv> r = p29_mul_scalar<Cv>(p29_from_std<Cv>(p), sc);
    xyzz_store<Cv>(data + (size_t)k * XZ, xyzz29_to_std<Cv>(r.v, r.inf));
}

// r_m = f_{d - m}, m <= d, zero up to M  (f: d + 1 coefficients)
template <class Cv>
__global__ __launch_bounds__(256) void k_ec_rev_coeffs(const uint32_t* __restrict__ f, uint32_t d, uint32_t M,
                                                       uint32_t* __restrict__ out) {
    using Fr = typename Cv::Fr;
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    fp_store<Fr>(out + (size_t)m * Fr::N, m <= d ? fp_load<Fr>(f + (size_t)(d - m) * Fr::N) : Fp<Fr>::zero());
}

// h_i = conv[d - 1 - i], i < d; identity up to n
template <class Cv>
__global__ __launch_bounds__(256) void k_ec_gather_h(const uint32_t* __restrict__ conv, uint32_t d, uint32_t n,
                                                     uint32_t* __restrict__ out) {
    constexpr int XZ = 4 * Cv::Fq::N;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    xyzz_store<Cv>(out + (size_t)i * XZ, i < d ? xyzz_load<Cv>(conv + (size_t)(d - 1 - i) * XZ) : XYZZ<Cv>::identity());
}

// XYZZ -> records of rec_words words: packed affine + u32 infinity flag
template <class Cv>
__global__ __launch_bounds__(EC_BLOCK) void k_ec_to_affine(const uint32_t* __restrict__ in, uint32_t count,
                                                           uint32_t rec_words, uint32_t* __restrict__ out) {
    constexpr int PT = 2 * Cv::Fq::N, XZ = 4 * Cv::Fq::N;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    XYZZ<Cv> p = xyzz_load<Cv>(in + (size_t)i * XZ);
    affine_store<Cv>(out + (size_t)i * rec_words, xyzz_to_affine<Cv>(p));
    out[(size_t)i * rec_words + PT] = p.is_inf() ? 1u : 0u;
}

// in-place group DFT of size 2^log_M, natural order in and out; inverse: with w^-1 and WITHOUT the 1 / M factor
template <class Cv>
static int32_t ec_ntt_t(int32_t curve, uint32_t* d_data, uint32_t* d_tmp, uint32_t* d_tw, uint32_t log_M, int inverse,
                        jfhip_stream* st) {
    constexpr int XZ = 4 * Cv::Fq::N;
    const uint32_t M = 1u << log_M;
    if (log_M == 0) return JFHIP_OK;
    const uint32_t *lo, *hi;
    uint32_t a;
    Lease lease;
    JF_TRY(ntt_root_tables(curve, log_M, &lo, &hi, &a, st, &lease));
    hipLaunchKernelGGL((k_ec_twiddles<Cv>), dim3((M / 2 + 255) / 256), dim3(256), 0, st->s, lo, hi, a, log_M, inverse, d_tw);
    hipLaunchKernelGGL((k_ec_bitrev<Cv>), dim3((M + 255) / 256), dim3(256), 0, st->s, d_data, log_M, d_tmp);
    JF_HIP(hipMemcpyAsync(d_data, d_tmp, (size_t)M * XZ * 4, hipMemcpyDeviceToDevice, st->s));
    for (uint32_t s = 1; s <= log_M; s++)
        hipLaunchKernelGGL((k_ec_layer<Cv>), dim3((M / 2 + EC_BLOCK - 1) / EC_BLOCK), dim3(EC_BLOCK), 0, st->s, d_data,
                           log_M, s, d_tw);
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

// (2^log_M)^-1 in Montgomery form
template <class Cv>
__global__ void k_ec_minv(uint32_t log_M, uint32_t* __restrict__ out) {
    using Fr = typename Cv::Fr;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const Fp<Fr> half = fp_inv<Fr>(fp_dbl<Fr>(Fp<Fr>::one()));
    fp_store<Fr>(out, fp_pow_u64<Fr>(half, log_M));
}

struct ScopedBuf {   // work buffer of one call: hipFree waits for the kernels that use it
    DevBuf b;
    ~ScopedBuf() { b.release(); }
};

template <class Cv>
static int32_t multi_open_rou_t(const jfhip_srs* srs, const uint32_t* d_coeffs, size_t ncoeffs, uint32_t log_dom,
                                size_t num_points, uint32_t* d_out, size_t rec_bytes, uint32_t log_M, jfhip_stream* st) {
    using Fr = typename Cv::Fr;
    constexpr int XZ = 4 * Cv::Fq::N;
    const uint32_t d = (uint32_t)(ncoeffs - 1);   // degree (>= 1 here)
    const uint32_t M = 1u << log_M, n = 1u << log_dom;
    const size_t big = (size_t)(M > n ? M : n);
    ScopedBuf pts, tmp, tw, sc, par;
    JF_TRY(pts.b.ensure(big * XZ * 4));
    JF_TRY(tmp.b.ensure(big * XZ * 4));
    JF_TRY(tw.b.ensure((big / 2 + 1) * Fr::N * 4));
    JF_TRY(sc.b.ensure((size_t)M * Fr::N * 4));
    JF_TRY(par.b.ensure(Fr::N * 4));
    uint32_t* P = pts.b.as<uint32_t>();
    uint32_t* T = tmp.b.as<uint32_t>();
    hipLaunchKernelGGL((k_ec_minv<Cv>), dim3(1), dim3(64), 0, st->s, log_M, par.b.as<uint32_t>());
    // DFT_M(S): S_k for k < d.  It depends on the SRS only: computed on the first multi-open of this shape, then copied
    // from the handle (a third of the group transforms of a call).  fk_mu covers the look-up and the publication only:
    // the transform itself runs unlocked, so commits and other opens against the handle are never held up by it.
    const size_t dft_bytes = (size_t)M * XZ * 4;
    const auto key = std::make_pair(d, log_M);
    bool have = false;
    {
        std::lock_guard<std::mutex> lk(srs->fk_mu);
        auto it = srs->fk_dft.find(key);
        if (it != srs->fk_dft.end()) {
            // enqueued under the lock: a release (hipFree, which waits for enqueued work) cannot overtake the copy
            JF_HIP(hipMemcpyAsync(P, it->second.p, dft_bytes, hipMemcpyDeviceToDevice, st->s));
            it->second.last = ++srs->fk_tick;
            have = true;
        }
    }
    if (!have) {
        hipLaunchKernelGGL((k_ec_from_srs<Cv>), dim3((M + 255) / 256), dim3(256), 0, st->s, srs->table, (int)srs->f29, d, M,
                           P);
        JF_TRY(ec_ntt_t<Cv>(srs->curve, P, T, tw.b.as<uint32_t>(), log_M, 0, st));
        uint32_t* keep = nullptr;
        if (dft_bytes > fk_cache_byte_cap()) {
            // larger than the whole budget (2^27 points are 16 / 24 GiB): recomputed per call, never pinned in HBM
        } else if (hipMalloc(&keep, dft_bytes) == hipSuccess) {
            // other host threads read it without any ordering on this stream: complete before it is published
            bool ok = hipMemcpyAsync(keep, P, dft_bytes, hipMemcpyDeviceToDevice, st->s) == hipSuccess &&
                      hipStreamSynchronize(st->s) == hipSuccess;
            std::vector<uint32_t*> drop;
            if (ok) {
                std::lock_guard<std::mutex> lk(srs->fk_mu);
                if (srs->fk_dft.count(key)) {
                    drop.push_back(keep);   // another thread published the same shape meanwhile
                } else {
                    while (srs->fk_bytes + dft_bytes > fk_cache_byte_cap() && !srs->fk_dft.empty()) {
                        auto victim = srs->fk_dft.begin();
                        for (auto j = srs->fk_dft.begin(); j != srs->fk_dft.end(); ++j)
                            if (j->second.last < victim->second.last) victim = j;
                        srs->fk_bytes -= victim->second.bytes;
                        drop.push_back(victim->second.p);
                        srs->fk_dft.erase(victim);
                    }
                    jfhip_srs::FkEntry e;
                    e.p = keep;
                    e.bytes = dft_bytes;
                    e.last = ++srs->fk_tick;
                    srs->fk_dft[key] = e;
                    srs->fk_bytes += dft_bytes;
                }
            } else {
                drop.push_back(keep);
            }
            for (uint32_t* q : drop) (void)hipFree(q);
        } else {
            (void)hipGetLastError();   // no room to keep it: recomputed next time
        }
    }
    // DFT_M(r), r_m = f_{d - m}
    hipLaunchKernelGGL((k_ec_rev_coeffs<Cv>), dim3((M + 255) / 256), dim3(256), 0, st->s, d_coeffs, d, M,
                       sc.b.as<uint32_t>());
    JF_TRY(ntt_device(srs->curve, sc.b.as<uint32_t>(), log_M, 1, 0, nullptr, 0, st));
    // pointwise product (with the 1 / M of the inverse transform), inverse group DFT: the convolution S * r
    hipLaunchKernelGGL((k_ec_scale<Cv>), dim3((M + EC_BLOCK - 1) / EC_BLOCK), dim3(EC_BLOCK), 0, st->s, P,
                       sc.b.as<uint32_t>(), par.b.as<uint32_t>(), M);
    JF_TRY(ec_ntt_t<Cv>(srs->curve, P, T, tw.b.as<uint32_t>(), log_M, 1, st));
    // h, zero-extended to the evaluation domain, and its group DFT: the proofs in the domain's order
    hipLaunchKernelGGL((k_ec_gather_h<Cv>), dim3((n + 255) / 256), dim3(256), 0, st->s, P, d, n, T);
    JF_HIP(hipMemcpyAsync(P, T, (size_t)n * XZ * 4, hipMemcpyDeviceToDevice, st->s));
    JF_TRY(ec_ntt_t<Cv>(srs->curve, P, T, tw.b.as<uint32_t>(), log_dom, 0, st));
    hipLaunchKernelGGL((k_ec_to_affine<Cv>), dim3((unsigned)((num_points + EC_BLOCK - 1) / EC_BLOCK)), dim3(EC_BLOCK), 0,
                       st->s, P, (uint32_t)num_points, (uint32_t)(rec_bytes / 4), d_out);
    JF_HIP(hipGetLastError());
    JF_HIP(hipStreamSynchronize(st->s));
    return JFHIP_OK;
}

// d_coeffs: ncoeffs >= 2 Fr coefficients (device); d_out: num_points <= 2^log_dom records of rec_bytes;
// 2^log_M >= 2 (ncoeffs - 1): the convolution length.  Synchronous: the results are complete on return.
int32_t multi_open_rou_device(const jfhip_srs* srs, const uint32_t* d_coeffs, size_t ncoeffs, uint32_t log_dom,
                              size_t num_points, uint32_t* d_out, size_t rec_bytes, uint32_t log_M, jfhip_stream* st) {
    if (srs->curve == JFHIP_BN254)
        return multi_open_rou_t<Bn254G1>(srs, d_coeffs, ncoeffs, log_dom, num_points, d_out, rec_bytes, log_M, st);
    return multi_open_rou_t<Bls381G1>(srs, d_coeffs, ncoeffs, log_dom, num_points, d_out, rec_bytes, log_M, st);
}

// ---- Lagrange-basis commit key -----------------------------------------------------------------------------------
// G_i = [L_i(beta)] g, i < n = 2^log_n, with L_i the Lagrange basis of the radix-2 domain {w^i}: since
// L_i(X) = (1 / n) sum_j w^(-ij) X^j, the points are the inverse group DFT of the monomial key S_j = [beta^j] g.  A
// polynomial given by its evaluations e_i on the domain then commits as sum_i e_i G_i -- the same group element as the
// monomial commitment of its coefficients -- and the prover's blinding (b_0 + b_1 X + ..)(X^n - 1) (prover.rs:457-466) adds
// sum_k b_k V_k with V_k = S_(n+k) - S_k, appended after the n Lagrange points.  out: n + blinders packed affine points.
template <class Cv>
__global__ __launch_bounds__(256) void k_ec_fill_one(uint32_t count, uint32_t* __restrict__ out) {
    using Fr = typename Cv::Fr;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) fp_store<Fr>(out + (size_t)k * Fr::N, Fp<Fr>::one());
}
// V_k = S_(n+k) - S_k for k < blinders, written after the n transformed points
template <class Cv>
__global__ __launch_bounds__(64) void k_ec_vanishing_points(const uint32_t* __restrict__ table, int f29, uint32_t n,
                                                            uint32_t blinders, uint32_t* __restrict__ data) {
    using Q = typename Cv::Fq;
    constexpr int PT = 2 * Q::N, XZ = 4 * Q::N;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= blinders) return;
    Affine<Cv> hi = affine_load<Cv>(table + (size_t)(n + k) * PT), lo = affine_load<Cv>(table + (size_t)k * PT);
    if (f29) {
        hi.x = fp_scale_down_r29<Q>(hi.x), hi.y = fp_scale_down_r29<Q>(hi.y);
        lo.x = fp_scale_down_r29<Q>(lo.x), lo.y = fp_scale_down_r29<Q>(lo.y);
    }
    XYZZ<Cv> acc = hi.is_inf() ? XYZZ<Cv>::identity() : xyzz_from_affine<Cv>(hi);
    if (!lo.is_inf()) {
        lo.y = fp_neg<Q>(lo.y);
        xyzz_add_affine<Cv>(acc, lo);
    }
    xyzz_store<Cv>(data + (size_t)(n + k) * XZ, acc);
}
// XYZZ -> packed affine points (identity = all zero), no flag word
template <class Cv>
__global__ __launch_bounds__(EC_BLOCK) void k_ec_to_packed(const uint32_t* __restrict__ in, uint32_t count,
                                                           uint32_t* __restrict__ out) {
    constexpr int PT = 2 * Cv::Fq::N, XZ = 4 * Cv::Fq::N;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    affine_store<Cv>(out + (size_t)i * PT, xyzz_to_affine<Cv>(xyzz_load<Cv>(in + (size_t)i * XZ)));
}

template <class Cv>
static int32_t srs_lagrange_points_t(const jfhip_srs* srs, uint32_t log_n, uint32_t blinders, uint32_t* d_packed,
                                     jfhip_stream* st) {
    using Fr = typename Cv::Fr;
    constexpr int XZ = 4 * Cv::Fq::N;
    const uint32_t n = 1u << log_n;
    const size_t total = (size_t)n + blinders;
    ScopedBuf pts, tmp, tw, ones, par;
    JF_TRY(pts.b.ensure(total * XZ * 4));
    JF_TRY(tmp.b.ensure((size_t)n * XZ * 4));
    JF_TRY(tw.b.ensure(((size_t)n / 2 + 1) * Fr::N * 4));
    JF_TRY(ones.b.ensure((size_t)n * Fr::N * 4));
    JF_TRY(par.b.ensure(Fr::N * 4));
    uint32_t* P = pts.b.as<uint32_t>();
    hipLaunchKernelGGL((k_ec_from_srs<Cv>), dim3((n + 255) / 256), dim3(256), 0, st->s, srs->table, (int)srs->f29, n, n, P);
    JF_TRY(ec_ntt_t<Cv>(srs->curve, P, tmp.b.as<uint32_t>(), tw.b.as<uint32_t>(), log_n, 1, st));
    // the 1 / n of the inverse transform
    hipLaunchKernelGGL((k_ec_minv<Cv>), dim3(1), dim3(64), 0, st->s, log_n, par.b.as<uint32_t>());
    hipLaunchKernelGGL((k_ec_fill_one<Cv>), dim3((n + 255) / 256), dim3(256), 0, st->s, n, ones.b.as<uint32_t>());
    if (log_n)
        hipLaunchKernelGGL((k_ec_scale<Cv>), dim3((n + EC_BLOCK - 1) / EC_BLOCK), dim3(EC_BLOCK), 0, st->s, P,
                           ones.b.as<uint32_t>(), par.b.as<uint32_t>(), n);
    if (blinders)
        hipLaunchKernelGGL((k_ec_vanishing_points<Cv>), dim3((blinders + 63) / 64), dim3(64), 0, st->s, srs->table,
                           (int)srs->f29, n, blinders, P);
    hipLaunchKernelGGL((k_ec_to_packed<Cv>), dim3((unsigned)((total + EC_BLOCK - 1) / EC_BLOCK)), dim3(EC_BLOCK), 0, st->s, P,
                       (uint32_t)total, d_packed);
    JF_HIP(hipGetLastError());
    JF_HIP(hipStreamSynchronize(st->s));   // the work buffers go out of scope
    return JFHIP_OK;
}

int32_t srs_lagrange_points(const jfhip_srs* srs, uint32_t log_n, uint32_t blinders, uint32_t* d_packed, jfhip_stream* st) {
    if (srs->curve == JFHIP_BN254) return srs_lagrange_points_t<Bn254G1>(srs, log_n, blinders, d_packed, st);
    return srs_lagrange_points_t<Bls381G1>(srs, log_n, blinders, d_packed, st);
}

// give back the FK23 transforms a handle keeps (jfhip_srs_release_caches, jfhip_cache_clear, jfhip_srs_free)
void srs_release_fk_cache(const jfhip_srs* srs) {
    std::vector<uint32_t*> drop;
    {
        std::lock_guard<std::mutex> lk(srs->fk_mu);
        for (auto& kv : srs->fk_dft) drop.push_back(kv.second.p);
        srs->fk_dft.clear();
        srs->fk_bytes = 0;
    }
    for (uint32_t* q : drop) (void)hipFree(q);
}

}  // namespace jf
