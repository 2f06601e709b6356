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

struct QuotDomain {
    uint32_t* inv_den = nullptr;  // 1 / (n (g w^i - 1)), i < m
    uint32_t* zh_inv = nullptr;   // 1 / ((g w^i)^n - 1), i < ratio
    size_t nbytes = 0;
    size_t bytes() const { return nbytes; }
    void release() {
        if (inv_den) (void)hipFree(inv_den);
        if (zh_inv) (void)hipFree(zh_inv);
        inv_den = zh_inv = nullptr;
    }
};
struct QuotKey {
    int device, curve;
    uint32_t log_n, log_m;
    uint32_t builtin_gen;  // Fr::GENERATOR as a small integer when no explicit offset was given, else 0
    uint32_t offset[8];    // explicit coset offset (Montgomery limbs)
    bool operator<(const QuotKey& o) const { return memcmp(this, &o, sizeof(QuotKey)) < 0; }
};
static DevCache<QuotKey, QuotDomain> g_qdom;
void quotient_cache_clear() { g_qdom.clear(); }

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

// coset offset of the evaluation domain as a kernel argument: Fr::GENERATOR (gen != 0) or an arbitrary element
// (a rank's slice h w_m^g of the quotient coset, see jfhip_plonk_quotient_coset_device)
struct CosetOffset {
    uint32_t gen;
    uint32_t v[8];
};
template <class Fr>
JF_DEV Fp<Fr> coset_offset(const CosetOffset& o) {
    static_assert(Fr::N == 8, "scalar fields have 8 limbs");
    if (o.gen) return fp_from_small<Fr>(o.gen);
    Fp<Fr> x;
#pragma unroll
    for (int i = 0; i < 8; i++) x.v[i] = o.v[i];
    return x;
}

constexpr int INV_CHUNK = 8;
// inv_den[i] = 1 / (n (g w^i - 1)): each lane inverts INV_CHUNK consecutive values with one inversion
template <class Fr>
__global__ __launch_bounds__(128) void k_quot_inv_den(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi,
                                                      uint32_t a, uint32_t log_n, uint32_t log_m, CosetOffset gen,
                                                      uint32_t* __restrict__ inv_den) {
    constexpr int N = Fr::N;
    const uint64_t m = 1ull << log_m;
    uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * INV_CHUNK;
    if (i0 >= m) return;
    const Fp<Fr> g = coset_offset<Fr>(gen);
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
    Fp<Fr> inv = fp_inv_safegcd<Fr>(prod);
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
                              uint32_t log_n, uint32_t ratio, CosetOffset gen, uint32_t* __restrict__ zh_inv) {
    uint32_t i = threadIdx.x;
    if (i >= ratio || blockIdx.x) return;
    Fp<Fr> x = fp_mul<Fr>(coset_offset<Fr>(gen), root_pow<Fr>(lo, hi, a, i));
    for (uint32_t k = 0; k < log_n; k++) x = fp_sqr<Fr>(x);
    x = fp_sub<Fr>(x, Fp<Fr>::one());
    fp_store<Fr>(zh_inv + (size_t)i * Fr::N, fp_inv_safegcd<Fr>(x));
}

// in: k[6], alpha, beta, gamma, tau, alpha_base  ->  derived constants (QC_* slots)
enum {
    QC_KB = 0,  // k_j * beta, 6 slots
    QC_ALPHA = 6, QC_ALPHA2, QC_BETA, QC_GAMMA, QC_ABASE, QC_ONE, QC_GEN, QC_TAU, QC_ALPHA3, QC_ALPHA4, QC_ALPHA5,
    QC_ALPHA6, QC_B1, QC_GB1, QC_WN_INV, QC_COUNT
};
constexpr int QUOT_NIN = 11;

template <class Fr>
__global__ void k_quot_consts(const uint32_t* __restrict__ in, CosetOffset gen, const uint32_t* __restrict__ lo,
                              const uint32_t* __restrict__ hi, uint32_t a, uint32_t log_n, uint32_t log_m,
                              uint32_t* __restrict__ out) {
    constexpr int N = Fr::N;
    if (threadIdx.x || blockIdx.x) return;
    Fp<Fr> alpha = fp_load<Fr>(in + 6 * N), beta = fp_load<Fr>(in + 7 * N), gamma = fp_load<Fr>(in + 8 * N);
    for (int j = 0; j < 6; j++) fp_store<Fr>(out + (QC_KB + j) * N, fp_mul<Fr>(fp_load<Fr>(in + j * N), beta));
    Fp<Fr> a2 = fp_sqr<Fr>(alpha), a3 = fp_mul<Fr>(a2, alpha), a4 = fp_mul<Fr>(a3, alpha), a5 = fp_mul<Fr>(a4, alpha);
    fp_store<Fr>(out + QC_ALPHA * N, alpha);
    fp_store<Fr>(out + QC_ALPHA2 * N, a2);
    fp_store<Fr>(out + QC_BETA * N, beta);
    fp_store<Fr>(out + QC_GAMMA * N, gamma);
    fp_store<Fr>(out + QC_ABASE * N, fp_load<Fr>(in + 10 * N));
    fp_store<Fr>(out + QC_ONE * N, Fp<Fr>::one());
    fp_store<Fr>(out + QC_GEN * N, coset_offset<Fr>(gen));
    fp_store<Fr>(out + QC_TAU * N, fp_load<Fr>(in + 9 * N));
    fp_store<Fr>(out + QC_ALPHA3 * N, a3);
    fp_store<Fr>(out + QC_ALPHA4 * N, a4);
    fp_store<Fr>(out + QC_ALPHA5 * N, a5);
    fp_store<Fr>(out + QC_ALPHA6 * N, fp_mul<Fr>(a5, alpha));
    Fp<Fr> b1 = fp_add<Fr>(Fp<Fr>::one(), beta);
    fp_store<Fr>(out + QC_B1 * N, b1);
    fp_store<Fr>(out + QC_GB1 * N, fp_mul<Fr>(b1, gamma));
    // w_n^-1 = w_m^(m - m/n)   (domain.group_gen_inv, prover.rs:871)
    const uint64_t m = 1ull << log_m;
    fp_store<Fr>(out + QC_WN_INV * N, root_pow<Fr>(lo, hi, a, m - (m >> log_n)));
}

// One point of the quotient.  NW = 5: TurboPlonk (13 selectors).  NW = 6: UltraPlonk (14 selectors, q_lookup
// last) plus the Plookup part (prover.rs:766-882); lk = [range_table, key_table, table_dom_sep, q_dom_sep,
// h1, h2, prod_lookup].  L1(x)/Z_H(x) = inv_den[i]; Ln(x)/Z_H(x) = w^-1 / (n (x - w^-1)) = inv_den[i + ratio]
// because x_i - w_n^-1 = w_n^-1 (x_(i+ratio) - 1)   (plonk/src/lagrange.rs:89-115).
template <class Fr, int NW>
__global__ __launch_bounds__(256) void k_quot(uint32_t log_n, uint32_t log_m, const uint32_t* __restrict__ sel,
                                              const uint32_t* __restrict__ sig, const uint32_t* __restrict__ wires,
                                              const uint32_t* __restrict__ zperm, const uint32_t* __restrict__ pi,
                                              const uint32_t* __restrict__ lk, const uint32_t* __restrict__ cst,
                                              const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi,
                                              uint32_t a, const uint32_t* __restrict__ inv_den,
                                              const uint32_t* __restrict__ zh_inv, uint32_t zmask,
                                              uint32_t* __restrict__ out) {
    // zmask bit j: selector polynomial j is identically zero (a gate kind the circuit never uses): its row is not read
    // and the products it would scale are not computed (wave-uniform branches); pi == nullptr: empty public input
    constexpr int N = Fr::N;
    constexpr bool ULTRA = NW == 6;
    using F = Fp<Fr>;
    const size_t m = (size_t)1 << log_m;
    const size_t ratio = m >> log_n;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const size_t nx = (i + ratio) & (m - 1);
    auto LD = [&](const uint32_t* base, int poly) { return fp_load<Fr>(base + ((size_t)poly * m + i) * N); };
    auto LDN = [&](const uint32_t* base, int poly) { return fp_load<Fr>(base + ((size_t)poly * m + nx) * N); };
    auto CS = [&](int idx) { return fp_load<Fr>(cst + idx * N); };

    F w[NW];
#pragma unroll
    for (int j = 0; j < NW; j++) w[j] = LD(wires, j);

    // ---- gate part (prover.rs:665-699); selector order q_lc[4] q_mul[2] q_hash[4] q_o q_c q_ecc
    auto live = [&](int j) { return ((zmask >> j) & 1u) == 0u; };
    F acc = Fp<Fr>::zero();
    if (live(11)) acc = LD(sel, 11);
    if (pi) acc = fp_add<Fr>(acc, fp_load<Fr>(pi + i * N));
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (live(j)) acc = fp_add<Fr>(acc, fp_mul<Fr>(LD(sel, j), w[j]));
    if (live(4) || live(5) || live(12)) {
        F w01 = fp_mul<Fr>(w[0], w[1]), w23 = fp_mul<Fr>(w[2], w[3]);
        if (live(4)) acc = fp_add<Fr>(acc, fp_mul<Fr>(LD(sel, 4), w01));
        if (live(5)) acc = fp_add<Fr>(acc, fp_mul<Fr>(LD(sel, 5), w23));
        if (live(12)) acc = fp_add<Fr>(acc, fp_mul<Fr>(LD(sel, 12), fp_mul<Fr>(fp_mul<Fr>(w01, w23), w[4])));
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (live(6 + j)) {
            F w2 = fp_sqr<Fr>(w[j]);
            F w5 = fp_mul<Fr>(fp_sqr<Fr>(w2), w[j]);
            acc = fp_add<Fr>(acc, fp_mul<Fr>(LD(sel, 6 + j), w5));
        }
    }
    if (live(10)) acc = fp_sub<Fr>(acc, fp_mul<Fr>(LD(sel, 10), w[4]));

    // ---- copy-constraint part (prover.rs:709-752)
    const F x = fp_mul<Fr>(CS(QC_GEN), root_pow<Fr>(lo, hi, a, i));  // eval_point = g * w^i
    const F beta = CS(QC_BETA), gamma = CS(QC_GAMMA);
    const F zx = fp_load<Fr>(zperm + i * N);
    const F zxw = fp_load<Fr>(zperm + nx * N);
    F r1 = zx, r1b = zxw;
#pragma unroll
    for (int j = 0; j < NW; j++) {
        F wg = fp_add<Fr>(w[j], gamma);
        r1 = fp_mul<Fr>(r1, fp_add<Fr>(wg, fp_mul<Fr>(CS(QC_KB + j), x)));
        r1b = fp_mul<Fr>(r1b, fp_add<Fr>(wg, fp_mul<Fr>(LD(sig, j), beta)));
    }
    const F l1d = fp_load<Fr>(inv_den + i * N);
    F t1 = fp_add<Fr>(acc, fp_mul<Fr>(CS(QC_ALPHA), fp_sub<Fr>(r1, r1b)));
    F t2 = fp_mul<Fr>(fp_mul<Fr>(CS(QC_ALPHA2), fp_sub<Fr>(zx, CS(QC_ONE))), l1d);

    if constexpr (ULTRA) {
        // ---- Plookup part (prover.rs:766-882)
        const F tau = CS(QC_TAU);
        const F lnd = fp_load<Fr>(inv_den + nx * N);
        const F ql = live(13) ? LD(sel, 13) : Fp<Fr>::zero(), qlw = live(13) ? LDN(sel, 13) : Fp<Fr>::zero();
        const F h1x = LD(lk, 4), h1w = LDN(lk, 4), h2x = LD(lk, 5), h2w = LDN(lk, 5), px = LD(lk, 6), pw = LDN(lk, 6);
        // eval_merged_table at x and x*w, eval_merged_lookup_witness at x (structs.rs:899-931)
        F t = fp_add<Fr>(fp_mul<Fr>(tau, w[4]), w[3]);
        t = fp_add<Fr>(fp_mul<Fr>(t, tau), LD(lk, 1));
        t = fp_add<Fr>(fp_mul<Fr>(t, tau), LD(lk, 2));
        const F mtx = fp_add<Fr>(fp_mul<Fr>(fp_mul<Fr>(t, tau), ql), LD(lk, 0));
        t = fp_add<Fr>(fp_mul<Fr>(tau, LDN(wires, 4)), LDN(wires, 3));
        t = fp_add<Fr>(fp_mul<Fr>(t, tau), LDN(lk, 1));
        t = fp_add<Fr>(fp_mul<Fr>(t, tau), LDN(lk, 2));
        const F mtw = fp_add<Fr>(fp_mul<Fr>(fp_mul<Fr>(t, tau), qlw), LDN(lk, 0));
        t = fp_add<Fr>(fp_mul<Fr>(tau, w[2]), w[1]);
        t = fp_add<Fr>(fp_mul<Fr>(t, tau), w[0]);
        t = fp_add<Fr>(fp_mul<Fr>(t, tau), LD(lk, 3));
        const F mlx = fp_add<Fr>(fp_mul<Fr>(fp_mul<Fr>(t, tau), ql), w[5]);
        // alpha^3 (h1(x) - h2(xw)) Ln/Z_H + alpha^4 (p(x) - 1) L1/Z_H + alpha^5 (p(x) - 1) Ln/Z_H
        const F pm1 = fp_sub<Fr>(px, CS(QC_ONE));
        F res2 = fp_mul<Fr>(CS(QC_ALPHA3), fp_mul<Fr>(fp_sub<Fr>(h1x, h2w), lnd));
        res2 = fp_add<Fr>(res2, fp_mul<Fr>(CS(QC_ALPHA4), fp_mul<Fr>(pm1, l1d)));
        res2 = fp_add<Fr>(res2, fp_mul<Fr>(CS(QC_ALPHA5), fp_mul<Fr>(pm1, lnd)));
        // alpha^6 (x - w^-1) [ p(x)(1+beta)(gamma + ml)(gb + mt(x) + beta mt(xw)) - p(xw)(gb + h1 + beta h1w)(gb + h2 + beta h2w) ]
        const F gb = CS(QC_GB1);
        F s1 = fp_mul<Fr>(fp_mul<Fr>(px, CS(QC_B1)), fp_add<Fr>(gamma, mlx));
        s1 = fp_mul<Fr>(s1, fp_add<Fr>(fp_add<Fr>(gb, mtx), fp_mul<Fr>(beta, mtw)));
        F s2 = fp_mul<Fr>(pw, fp_add<Fr>(fp_add<Fr>(gb, h1x), fp_mul<Fr>(beta, h1w)));
        s2 = fp_mul<Fr>(s2, fp_add<Fr>(fp_add<Fr>(gb, h2x), fp_mul<Fr>(beta, h2w)));
        F p3 = fp_mul<Fr>(fp_sub<Fr>(x, CS(QC_WN_INV)), fp_sub<Fr>(s1, s2));
        t1 = fp_add<Fr>(t1, fp_mul<Fr>(CS(QC_ALPHA6), p3));
        t2 = fp_add<Fr>(t2, res2);
    }
    // ---- combine (prover.rs:636-641, 644-655)
    F t = fp_add<Fr>(fp_mul<Fr>(t1, fp_load<Fr>(zh_inv + (i & (ratio - 1)) * N)), t2);
    F o = fp_load<Fr>(out + i * N);
    fp_store<Fr>(out + i * N, fp_add<Fr>(o, fp_mul<Fr>(CS(QC_ABASE), t)));
}

template <class Fr>
static int32_t quotient_t(int32_t curve, int nw, uint32_t log_n, uint32_t log_m, const uint32_t* sel,
                          const uint32_t* sig, const uint32_t* wires, const uint32_t* zperm, const uint32_t* pi,
                          const uint32_t* lookup, const uint64_t* k_mont, const uint64_t* ch, uint32_t* out,
                          jfhip_stream* st, const uint64_t* offset_mont = nullptr, uint32_t zmask = 0) {
    constexpr int N = Fr::N;
    CosetOffset gen;
    memset(&gen, 0, sizeof(gen));
    if (offset_mont) memcpy(gen.v, offset_mont, 32);
    else gen.gen = curve == JFHIP_BN254 ? 5u : 7u;  // Fr::GENERATOR
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
    if (offset_mont) {
        bool zero = true;
        for (int i = 0; i < 4; i++) zero = zero && offset_mont[i] == 0;
        if (zero) {
            set_error("quotient: coset offset 0 is not a coset");
            return JFHIP_ERR_INVALID;
        }
    }
    const uint32_t *lo, *hi;
    uint32_t a;
    Lease root_lease, dom_lease;  // held until the kernels below are enqueued
    JF_TRY(ntt_root_tables(curve, log_m, &lo, &hi, &a, st, &root_lease));
    QuotDomain qd;
    {
        QuotKey key;
        memset(&key, 0, sizeof(key));
        JF_HIP(hipGetDevice(&key.device));
        key.curve = curve;
        key.log_n = log_n;
        key.log_m = log_m;
        key.builtin_gen = gen.gen;
        memcpy(key.offset, gen.v, 32);
        JF_TRY(g_qdom.get(key, &qd, &dom_lease, m * N * 4, [&](QuotDomain* q) -> int32_t {
            q->nbytes = (m + ratio) * N * 4;
            JF_HIP(hipMalloc(&q->inv_den, m * N * 4));
            JF_HIP(hipMalloc(&q->zh_inv, (size_t)ratio * N * 4));
            const uint64_t lanes = (m + INV_CHUNK - 1) / INV_CHUNK;
            hipLaunchKernelGGL((k_quot_inv_den<Fr>), dim3((unsigned)((lanes + 127) / 128)), dim3(128), 0, st->s, lo,
                               hi, a, log_n, log_m, gen, q->inv_den);
            hipLaunchKernelGGL((k_quot_zh_inv<Fr>), dim3(1), dim3(64), 0, st->s, lo, hi, a, log_n, ratio, gen,
                               q->zh_inv);
            JF_HIP(hipGetLastError());
            JF_HIP(hipStreamSynchronize(st->s));
            return JFHIP_OK;
        }));
    }
    // host -> device: k[6] (padded), alpha, beta, gamma, tau, alpha_base
    JF_TRY(st->quot_consts.ensure((QUOT_NIN + QC_COUNT) * N * 4));
    uint32_t* d_in = st->quot_consts.as<uint32_t>();
    uint32_t* d_cst = d_in + QUOT_NIN * N;
    uint64_t host_in[QUOT_NIN * 4];
    memset(host_in, 0, sizeof(host_in));
    memcpy(host_in, k_mont, (size_t)nw * 32);
    if (nw == 5) {  // ch = alpha, beta, gamma, alpha_base
        memcpy(host_in + 6 * 4, ch, 3 * 32);
        memcpy(host_in + 10 * 4, ch + 3 * 4, 32);
    } else {  // ch = alpha, beta, gamma, tau, alpha_base
        memcpy(host_in + 6 * 4, ch, 5 * 32);
    }
    // pageable source: the copy has read host_in when the call returns
    JF_HIP(hipMemcpyAsync(d_in, host_in, sizeof(host_in), hipMemcpyHostToDevice, st->s));
    hipLaunchKernelGGL((k_quot_consts<Fr>), dim3(1), dim3(64), 0, st->s, d_in, gen, lo, hi, a, log_n, log_m, d_cst);
    const dim3 grid((unsigned)((m + 255) / 256));
    if (nw == 5)
        hipLaunchKernelGGL((k_quot<Fr, 5>), grid, dim3(256), 0, st->s, log_n, log_m, sel, sig, wires, zperm, pi,
                           (const uint32_t*)nullptr, d_cst, lo, hi, a, qd.inv_den, qd.zh_inv, zmask, out);
    else
        hipLaunchKernelGGL((k_quot<Fr, 6>), grid, dim3(256), 0, st->s, log_n, log_m, sel, sig, wires, zperm, pi, lookup,
                           d_cst, lo, hi, a, qd.inv_den, qd.zh_inv, zmask, out);
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

int32_t plonk_quotient_turbo_device(int32_t curve, uint32_t log_n, uint32_t log_m, const uint32_t* sel,
                                    const uint32_t* sig, const uint32_t* wires, const uint32_t* zperm,
                                    const uint32_t* pi, const uint64_t* k_mont, const uint64_t* challenges_mont,
                                    uint32_t* out, jfhip_stream* st) {
    if (curve == JFHIP_BN254)
        return quotient_t<Bn254Fr>(curve, 5, log_n, log_m, sel, sig, wires, zperm, pi, nullptr, k_mont, challenges_mont,
                                   out, st);
    return quotient_t<Bls381Fr>(curve, 5, log_n, log_m, sel, sig, wires, zperm, pi, nullptr, k_mont, challenges_mont, out,
                                st);
}

int32_t plonk_quotient_coset_device(int32_t curve, int nw, uint32_t log_n, uint32_t log_m, const uint64_t* offset_mont,
                                    const uint32_t* sel, const uint32_t* sig, const uint32_t* wires,
                                    const uint32_t* zperm, const uint32_t* pi, const uint32_t* lookup,
                                    const uint64_t* k_mont, const uint64_t* challenges_mont, uint32_t* out,
                                    jfhip_stream* st, uint32_t zmask) {
    if (curve == JFHIP_BN254)
        return quotient_t<Bn254Fr>(curve, nw, log_n, log_m, sel, sig, wires, zperm, pi, lookup, k_mont, challenges_mont,
                                   out, st, offset_mont, zmask);
    return quotient_t<Bls381Fr>(curve, nw, log_n, log_m, sel, sig, wires, zperm, pi, lookup, k_mont, challenges_mont, out,
                                st, offset_mont, zmask);
}

int32_t plonk_quotient_ultra_device(int32_t curve, uint32_t log_n, uint32_t log_m, const uint32_t* sel,
                                    const uint32_t* sig, const uint32_t* wires, const uint32_t* zperm,
                                    const uint32_t* pi, const uint32_t* lookup, const uint64_t* k_mont,
                                    const uint64_t* challenges_mont, uint32_t* out, jfhip_stream* st) {
    if (curve == JFHIP_BN254)
        return quotient_t<Bn254Fr>(curve, 6, log_n, log_m, sel, sig, wires, zperm, pi, lookup, k_mont, challenges_mont,
                                   out, st);
    return quotient_t<Bls381Fr>(curve, 6, log_n, log_m, sel, sig, wires, zperm, pi, lookup, k_mont, challenges_mont, out,
                                st);
}

}  // namespace jf
