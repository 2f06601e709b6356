// TurboPlonk quotient-polynomial evaluations on the device: the per-point loop
// of compute_quotient_polynomial, reference plonk/src/proof_system/prover.rs:587-655 with the gate
// part :665-699 and the copy-constraint part :709-752, one instance, no Plookup:
//     t_i = (t_circ + t_perm_1) * z_h_inv[i mod ratio] + t_perm_2,     out[i] += alpha_base * t_i
// Inputs are the coset evaluations (offset Fr::GENERATOR, size m = ratio * n) the prover already
// holds on the device after its coset NTTs, laid out [poly][m].  Without this kernel a 2^20-gate
// proof moves >= 6 GiB of evaluations over PCIe for the host loop.
//
// HBM-bound by design: 26 loads + 1 load/store of 32 B per point.  The reference's per-point field
// division (:748-749) is replaced by a per-domain table 1/(n (x_i - 1)) built once with Montgomery
// batch inversion; exact field arithmetic makes the result identical.
#include <map>
#include <mutex>

#include "common.hpp"
#include "fp.hpp"

namespace jf {

constexpr int QUOT_NSEL = 13;
constexpr int QUOT_NW = 5;
// derived constants, 8 limbs each: [0..5) k_j*beta, 5 alpha, 6 alpha^2, 7 beta, 8 gamma, 9 alpha_base,
// 10 one, 11 generator g
constexpr int QUOT_NCONST = 12;

struct QuotDomain {
    uint32_t* inv_den = nullptr;  // 1 / (n (g w^i - 1)), i < m
    uint32_t* zh_inv = nullptr;   // 1 / ((g w^i)^n - 1), i < ratio
};
struct QuotKey {
    int device, curve;
    uint32_t log_n, log_m;
    bool operator<(const QuotKey& o) const { return memcmp(this, &o, sizeof(QuotKey)) < 0; }
};
static std::mutex g_q_mu;
static std::map<QuotKey, QuotDomain> g_qdom;

template <class Fr>
JF_DEV Fp<Fr> root_pow(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi, uint32_t a, uint64_t e) {
    Fp<Fr> x = fp_load<Fr>(lo + (size_t)(e & ((1ull << a) - 1)) * Fr::N);
    uint64_t h = e >> a;
    if (h) x = fp_mul<Fr>(x, fp_load<Fr>(hi + (size_t)h * Fr::N));
    return x;
}

template <class Fr>
JF_DEV Fp<Fr> fp_from_small(uint32_t v) {
    Fp<Fr> t = Fp<Fr>::zero();
    t.v[0] = v;
    return fp_to_mont<Fr>(t);
}

constexpr int INV_CHUNK = 8;
// inv_den[i] = 1 / (n (g w^i - 1)): each lane inverts INV_CHUNK consecutive values with one inversion
template <class Fr>
__global__ __launch_bounds__(128) void k_quot_inv_den(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi,
                                                      uint32_t a, uint32_t log_n, uint32_t log_m, uint32_t gen,
                                                      uint32_t* __restrict__ inv_den) {
    constexpr int N = Fr::N;
    const uint64_t m = 1ull << log_m;
    uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * INV_CHUNK;
    if (i0 >= m) return;
    const Fp<Fr> g = fp_from_small<Fr>(gen);
    const Fp<Fr> one = Fp<Fr>::one();
    // n in Montgomery form: 2^log_n
    Fp<Fr> nn = one;
    for (uint32_t k = 0; k < log_n; k++) nn = fp_dbl<Fr>(nn);
    Fp<Fr> den[INV_CHUNK], pre[INV_CHUNK];
    Fp<Fr> prod = one;
#pragma unroll
    for (int k = 0; k < INV_CHUNK; k++) {
        Fp<Fr> x = fp_mul<Fr>(g, root_pow<Fr>(lo, hi, a, i0 + k));
        den[k] = fp_mul<Fr>(nn, fp_sub<Fr>(x, one));  // never zero: the coset avoids H
        pre[k] = prod;
        prod = fp_mul<Fr>(prod, den[k]);
    }
    Fp<Fr> inv = fp_inv_binary<Fr>(prod);
#pragma unroll
    for (int k = INV_CHUNK - 1; k >= 0; k--) {
        Fp<Fr> r = fp_mul<Fr>(inv, pre[k]);
        inv = fp_mul<Fr>(inv, den[k]);
        if (i0 + k < m) fp_store<Fr>(inv_den + (size_t)(i0 + k) * N, r);
    }
}

// z_h_inv[i] = 1 / ((g w^i)^n - 1), i < ratio   (prover.rs:509-516)
template <class Fr>
__global__ void k_quot_zh_inv(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi, uint32_t a,
                              uint32_t log_n, uint32_t ratio, uint32_t gen, uint32_t* __restrict__ zh_inv) {
    uint32_t i = threadIdx.x;
    if (i >= ratio || blockIdx.x) return;
    Fp<Fr> x = fp_mul<Fr>(fp_from_small<Fr>(gen), root_pow<Fr>(lo, hi, a, i));
    for (uint32_t k = 0; k < log_n; k++) x = fp_sqr<Fr>(x);
    x = fp_sub<Fr>(x, Fp<Fr>::one());
    fp_store<Fr>(zh_inv + (size_t)i * Fr::N, fp_inv_binary<Fr>(x));
}

// in: k[5], alpha, beta, gamma, alpha_base  ->  derived constants
template <class Fr>
__global__ void k_quot_consts(const uint32_t* __restrict__ in, uint32_t gen, uint32_t* __restrict__ out) {
    constexpr int N = Fr::N;
    if (threadIdx.x || blockIdx.x) return;
    Fp<Fr> alpha = fp_load<Fr>(in + 5 * N), beta = fp_load<Fr>(in + 6 * N);
    for (int j = 0; j < QUOT_NW; j++) fp_store<Fr>(out + j * N, fp_mul<Fr>(fp_load<Fr>(in + j * N), beta));
    fp_store<Fr>(out + 5 * N, alpha);
    fp_store<Fr>(out + 6 * N, fp_sqr<Fr>(alpha));
    fp_store<Fr>(out + 7 * N, beta);
    fp_store<Fr>(out + 8 * N, fp_load<Fr>(in + 7 * N));
    fp_store<Fr>(out + 9 * N, fp_load<Fr>(in + 8 * N));
    fp_store<Fr>(out + 10 * N, Fp<Fr>::one());
    fp_store<Fr>(out + 11 * N, fp_from_small<Fr>(gen));
}

template <class Fr>
__global__ __launch_bounds__(256) void k_quot_turbo(uint32_t log_n, uint32_t log_m, const uint32_t* __restrict__ sel,
                                                    const uint32_t* __restrict__ sig,
                                                    const uint32_t* __restrict__ wires,
                                                    const uint32_t* __restrict__ zperm,
                                                    const uint32_t* __restrict__ pi, const uint32_t* __restrict__ cst,
                                                    const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi,
                                                    uint32_t a, const uint32_t* __restrict__ inv_den,
                                                    const uint32_t* __restrict__ zh_inv, uint32_t* __restrict__ out) {
    constexpr int N = Fr::N;
    using F = Fp<Fr>;
    const size_t m = (size_t)1 << log_m;
    const size_t ratio = m >> log_n;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    auto LD = [&](const uint32_t* base, int poly) { return fp_load<Fr>(base + ((size_t)poly * m + i) * N); };
    auto CS = [&](int idx) { return fp_load<Fr>(cst + idx * N); };

    F w[QUOT_NW];
#pragma unroll
    for (int j = 0; j < QUOT_NW; j++) w[j] = LD(wires, j);

    // ---- gate part (prover.rs:665-699); selector order q_lc[4] q_mul[2] q_hash[4] q_o q_c q_ecc
    F acc = fp_add<Fr>(LD(sel, 11), fp_load<Fr>(pi + i * N));
#pragma unroll
    for (int j = 0; j < 4; j++) acc = fp_add<Fr>(acc, fp_mul<Fr>(LD(sel, j), w[j]));
    F w01 = fp_mul<Fr>(w[0], w[1]), w23 = fp_mul<Fr>(w[2], w[3]);
    acc = fp_add<Fr>(acc, fp_mul<Fr>(LD(sel, 4), w01));
    acc = fp_add<Fr>(acc, fp_mul<Fr>(LD(sel, 5), w23));
    acc = fp_add<Fr>(acc, fp_mul<Fr>(LD(sel, 12), fp_mul<Fr>(fp_mul<Fr>(w01, w23), w[4])));
#pragma unroll
    for (int j = 0; j < 4; j++) {
        F w2 = fp_sqr<Fr>(w[j]);
        F w5 = fp_mul<Fr>(fp_sqr<Fr>(w2), w[j]);
        acc = fp_add<Fr>(acc, fp_mul<Fr>(LD(sel, 6 + j), w5));
    }
    acc = fp_sub<Fr>(acc, fp_mul<Fr>(LD(sel, 10), w[4]));

    // ---- copy-constraint part (prover.rs:709-752)
    const F x = fp_mul<Fr>(CS(11), root_pow<Fr>(lo, hi, a, i));  // eval_point = g * w^i
    const F beta = CS(7), gamma = CS(8);
    const F zx = fp_load<Fr>(zperm + i * N);
    const F zxw = fp_load<Fr>(zperm + ((i + ratio) & (m - 1)) * N);
    F r1 = zx, r1b = zxw;
#pragma unroll
    for (int j = 0; j < QUOT_NW; j++) {
        F wg = fp_add<Fr>(w[j], gamma);
        r1 = fp_mul<Fr>(r1, fp_add<Fr>(wg, fp_mul<Fr>(CS(j), x)));
        r1b = fp_mul<Fr>(r1b, fp_add<Fr>(wg, fp_mul<Fr>(LD(sig, j), beta)));
    }
    F t1 = fp_add<Fr>(acc, fp_mul<Fr>(CS(5), fp_sub<Fr>(r1, r1b)));
    F t2 = fp_mul<Fr>(fp_mul<Fr>(CS(6), fp_sub<Fr>(zx, CS(10))), fp_load<Fr>(inv_den + i * N));
    // ---- combine (prover.rs:636-641, 644-655)
    F t = fp_add<Fr>(fp_mul<Fr>(t1, fp_load<Fr>(zh_inv + (i & (ratio - 1)) * N)), t2);
    F o = fp_load<Fr>(out + i * N);
    fp_store<Fr>(out + i * N, fp_add<Fr>(o, fp_mul<Fr>(CS(9), t)));
}

template <class Fr>
static int32_t quotient_turbo_t(int32_t curve, uint32_t log_n, uint32_t log_m, const uint32_t* sel,
                                const uint32_t* sig, const uint32_t* wires, const uint32_t* zperm, const uint32_t* pi,
                                const uint64_t* k_mont, const uint64_t* ch, uint32_t* out, jfhip_stream* st) {
    constexpr int N = Fr::N;
    const uint32_t gen = curve == JFHIP_BN254 ? 5u : 7u;  // Fr::GENERATOR
    if (log_m < log_n || log_m > 30) {
        set_error("quotient: bad domain sizes (log_n %u, log_m %u)", log_n, log_m);
        return JFHIP_ERR_INVALID;
    }
    const size_t m = (size_t)1 << log_m;
    const uint32_t ratio = 1u << (log_m - log_n);
    if (ratio > 64) {
        set_error("quotient: domain size ratio %u unsupported", ratio);
        return JFHIP_ERR_INVALID;
    }
    const uint32_t *lo, *hi;
    uint32_t a;
    JF_TRY(ntt_root_tables(curve, log_m, &lo, &hi, &a, st));
    QuotDomain qd;
    {
        QuotKey key;
        memset(&key, 0, sizeof(key));
        JF_HIP(hipGetDevice(&key.device));
        key.curve = curve;
        key.log_n = log_n;
        key.log_m = log_m;
        std::lock_guard<std::mutex> lk(g_q_mu);
        auto it = g_qdom.find(key);
        if (it == g_qdom.end()) {
            JF_HIP(hipMalloc(&qd.inv_den, m * N * 4));
            JF_HIP(hipMalloc(&qd.zh_inv, (size_t)ratio * N * 4));
            const uint64_t lanes = (m + INV_CHUNK - 1) / INV_CHUNK;
            hipLaunchKernelGGL((k_quot_inv_den<Fr>), dim3((unsigned)((lanes + 127) / 128)), dim3(128), 0, st->s, lo,
                               hi, a, log_n, log_m, gen, qd.inv_den);
            hipLaunchKernelGGL((k_quot_zh_inv<Fr>), dim3(1), dim3(64), 0, st->s, lo, hi, a, log_n, ratio, gen,
                               qd.zh_inv);
            JF_HIP(hipGetLastError());
            JF_HIP(hipStreamSynchronize(st->s));
            g_qdom[key] = qd;
        } else qd = it->second;
    }
    JF_TRY(st->quot_consts.ensure((9 + QUOT_NCONST) * N * 4));
    uint32_t* d_in = st->quot_consts.as<uint32_t>();
    uint32_t* d_cst = d_in + 9 * N;
    JF_HIP(hipMemcpyAsync(d_in, k_mont, QUOT_NW * 32, hipMemcpyHostToDevice, st->s));
    JF_HIP(hipMemcpyAsync(d_in + QUOT_NW * N, ch, 4 * 32, hipMemcpyHostToDevice, st->s));
    hipLaunchKernelGGL((k_quot_consts<Fr>), dim3(1), dim3(64), 0, st->s, d_in, gen, d_cst);
    hipLaunchKernelGGL((k_quot_turbo<Fr>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st->s, log_n, log_m, sel,
                       sig, wires, zperm, pi, d_cst, lo, hi, a, qd.inv_den, qd.zh_inv, out);
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

int32_t plonk_quotient_turbo_device(int32_t curve, uint32_t log_n, uint32_t log_m, const uint32_t* sel,
                                    const uint32_t* sig, const uint32_t* wires, const uint32_t* zperm,
                                    const uint32_t* pi, const uint64_t* k_mont, const uint64_t* challenges_mont,
                                    uint32_t* out, jfhip_stream* st) {
    if (curve == JFHIP_BN254)
        return quotient_turbo_t<Bn254Fr>(curve, log_n, log_m, sel, sig, wires, zperm, pi, k_mont, challenges_mont, out,
                                         st);
    return quotient_turbo_t<Bls381Fr>(curve, log_n, log_m, sel, sig, wires, zperm, pi, k_mont, challenges_mont, out, st);
}

}  // namespace jf
