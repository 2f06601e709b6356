// Grand-product builders of the PLONK prover on the device:
//   * compute_prod_permutation_polynomial  relation/src/constraint_system.rs:1394-1420
//       z[0] = 1,  z[j+1] = z[j] * prod_i (w_ij + gamma + beta id_ij) / prod_i (w_ij + gamma + beta sigma_ij)
//   * compute_lookup_prod_polynomial       relation/src/constraint_system.rs:1516-1575
//       p[0] = 1,  p[j+1] = p[j] * (1+beta)(gamma + f_j)(gamma(1+beta) + t_j + beta t_{j+1})
//                                 / ((gamma(1+beta) + s_j + beta s_{j+1})(gamma(1+beta) + s_{n-1+j} + beta s_{n+j}))
//       for j < n - 2, and p[n-1] = 1
// (values on H; the caller runs the size-n iNTT afterwards, as the reference does at :1418 / :1573).
// The reference loops serially with one field division per row -- the CPU-side cost left over once
// MSM and NTT are offloaded.  Here: per-row numerators / denominators, Montgomery batch inversion in
// chunks of 8 rows per lane, then a parallel prefix product (chunk products, recursive carries,
// second sweep).  Field arithmetic is exact, so the values equal the serial loop's bit for bit.
#include "common.hpp"
#include "fp.hpp"

namespace jf {

// a scalar-field element passed to a kernel by value
struct FpArg {
    uint32_t v[8];
};

constexpr int GP_INV_CHUNK = 8;
constexpr uint32_t GP_K = 32;  // prefix-product chunk per lane

// x[0] = 1, x[j+1] = a_j / b_j for j < n - 1
template <class Fr>
__global__ __launch_bounds__(128) void k_gp_perm_ratio(uint32_t n, uint32_t nw, const uint32_t* __restrict__ wires,
                                                       const uint32_t* __restrict__ idp,
                                                       const uint32_t* __restrict__ sig,
                                                       const uint32_t* __restrict__ cst, uint32_t* __restrict__ x) {
    constexpr int N = Fr::N;
    const uint64_t j0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * GP_INV_CHUNK;
    if (j0 + 1 >= n && j0 != 0) return;
    const Fp<Fr> beta = fp_load<Fr>(cst), gamma = fp_load<Fr>(cst + N);
    if (j0 == 0) fp_store<Fr>(x, Fp<Fr>::one());
    Fp<Fr> num[GP_INV_CHUNK], den[GP_INV_CHUNK], pre[GP_INV_CHUNK];
    Fp<Fr> prod = Fp<Fr>::one();
#pragma unroll
    for (int k = 0; k < GP_INV_CHUNK; k++) {
        const uint64_t j = j0 + k;
        Fp<Fr> a = Fp<Fr>::one(), b = Fp<Fr>::one();
        if (j + 1 < n) {
            for (uint32_t i = 0; i < nw; i++) {
                const size_t e = ((size_t)i * n + j) * N;
                Fp<Fr> t = fp_add<Fr>(fp_load<Fr>(wires + e), gamma);
                a = fp_mul<Fr>(a, fp_add<Fr>(t, fp_mul<Fr>(beta, fp_load<Fr>(idp + e))));
                b = fp_mul<Fr>(b, fp_add<Fr>(t, fp_mul<Fr>(beta, fp_load<Fr>(sig + e))));
            }
        }
        num[k] = a;
        den[k] = b;
        pre[k] = prod;
        prod = fp_mul<Fr>(prod, b);
    }
    Fp<Fr> inv = fp_inv_safegcd<Fr>(prod);  // a zero denominator (probability ~ n/r) gives 0, as F::inverse-less division would panic
#pragma unroll
    for (int k = GP_INV_CHUNK - 1; k >= 0; k--) {
        const uint64_t j = j0 + k;
        Fp<Fr> r = fp_mul<Fr>(num[k], fp_mul<Fr>(inv, pre[k]));
        inv = fp_mul<Fr>(inv, den[k]);
        if (j + 1 < n) fp_store<Fr>(x + (size_t)(j + 1) * N, r);
    }
}

// x[0] = 1, x[j+1] = a_j / b_j for j < n - 2   (x has n - 1 entries to scan; entry n-1 is set to 1 afterwards)
template <class Fr>
__global__ __launch_bounds__(128) void k_gp_lookup_ratio(uint32_t n, const uint32_t* __restrict__ f,
                                                         const uint32_t* __restrict__ t,
                                                         const uint32_t* __restrict__ s,
                                                         const uint32_t* __restrict__ cst, uint32_t* __restrict__ x) {
    constexpr int N = Fr::N;
    const uint64_t j0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * GP_INV_CHUNK;
    if (j0 + 2 >= n && j0 != 0) return;
    const Fp<Fr> beta = fp_load<Fr>(cst), gamma = fp_load<Fr>(cst + N);
    const Fp<Fr> b1 = fp_add<Fr>(Fp<Fr>::one(), beta), gb = fp_mul<Fr>(gamma, b1);
    if (j0 == 0) fp_store<Fr>(x, Fp<Fr>::one());
    Fp<Fr> num[GP_INV_CHUNK], den[GP_INV_CHUNK], pre[GP_INV_CHUNK];
    Fp<Fr> prod = Fp<Fr>::one();
    auto LD = [&](const uint32_t* p, uint64_t i) { return fp_load<Fr>(p + (size_t)i * N); };
#pragma unroll
    for (int k = 0; k < GP_INV_CHUNK; k++) {
        const uint64_t j = j0 + k;
        Fp<Fr> a = Fp<Fr>::one(), b = Fp<Fr>::one();
        if (j + 2 < n) {
            a = fp_mul<Fr>(b1, fp_add<Fr>(gamma, LD(f, j)));
            a = fp_mul<Fr>(a, fp_add<Fr>(fp_add<Fr>(gb, LD(t, j)), fp_mul<Fr>(beta, LD(t, j + 1))));
            b = fp_add<Fr>(fp_add<Fr>(gb, LD(s, j)), fp_mul<Fr>(beta, LD(s, j + 1)));
            b = fp_mul<Fr>(b, fp_add<Fr>(fp_add<Fr>(gb, LD(s, n - 1 + j)), fp_mul<Fr>(beta, LD(s, n + j))));
        }
        num[k] = a;
        den[k] = b;
        pre[k] = prod;
        prod = fp_mul<Fr>(prod, b);
    }
    Fp<Fr> inv = fp_inv_safegcd<Fr>(prod);
#pragma unroll
    for (int k = GP_INV_CHUNK - 1; k >= 0; k--) {
        const uint64_t j = j0 + k;
        Fp<Fr> r = fp_mul<Fr>(num[k], fp_mul<Fr>(inv, pre[k]));
        inv = fp_mul<Fr>(inv, den[k]);
        if (j + 2 < n) fp_store<Fr>(x + (size_t)(j + 1) * N, r);
    }
}

// ---- inclusive prefix product, in place ----------------------------------------------------------
// tot[c] = prod in[cK .. cK + K)
template <class Fr>
__global__ __launch_bounds__(128) void k_pp_chunks(const uint32_t* __restrict__ in, uint32_t m, uint32_t* __restrict__ tot) {
    constexpr int N = Fr::N;
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t lo = (uint64_t)c * GP_K;
    if (lo >= m) return;
    uint32_t hi = (uint32_t)((lo + GP_K < m) ? lo + GP_K : m);
    Fp<Fr> acc = fp_load<Fr>(in + (size_t)lo * N);
    for (uint32_t k = (uint32_t)lo + 1; k < hi; k++) acc = fp_mul<Fr>(acc, fp_load<Fr>(in + (size_t)k * N));
    fp_store<Fr>(tot + (size_t)c * N, acc);
}
// io[k] = carry * prod io[cK .. k], carry = upper[c - 1] (inclusive prefix of the chunk totals)
template <class Fr>
__global__ __launch_bounds__(128) void k_pp_apply(uint32_t* __restrict__ io, uint32_t m, const uint32_t* __restrict__ upper) {
    constexpr int N = Fr::N;
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t lo = (uint64_t)c * GP_K;
    if (lo >= m) return;
    uint32_t hi = (uint32_t)((lo + GP_K < m) ? lo + GP_K : m);
    Fp<Fr> acc = (upper && c > 0) ? fp_load<Fr>(upper + (size_t)(c - 1) * N) : Fp<Fr>::one();
    for (uint32_t k = (uint32_t)lo; k < hi; k++) {
        acc = fp_mul<Fr>(acc, fp_load<Fr>(io + (size_t)k * N));
        fp_store<Fr>(io + (size_t)k * N, acc);
    }
}
template <class Fr>
__global__ void k_gp_set_one(uint32_t* __restrict__ p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) fp_store<Fr>(p, Fp<Fr>::one());
}

template <class Fr>
static int32_t prefix_product_t(uint32_t* d_io, size_t m, jfhip_stream* st) {
    constexpr int N = Fr::N;
    std::vector<size_t> sizes;
    sizes.push_back(m);
    while (sizes.back() > GP_K) sizes.push_back((sizes.back() + GP_K - 1) / GP_K);
    const uint32_t levels = (uint32_t)sizes.size();
    size_t extra = 0;
    for (uint32_t l = 1; l < levels; l++) extra += sizes[l];
    JF_TRY(st->poly_b.ensure((extra + 1) * N * 4));
    std::vector<uint32_t*> buf(levels);
    buf[0] = d_io;
    size_t o = 0;
    for (uint32_t l = 1; l < levels; l++) {
        buf[l] = st->poly_b.as<uint32_t>() + o * N;
        o += sizes[l];
    }
    for (uint32_t l = 1; l < levels; l++)
        hipLaunchKernelGGL((k_pp_chunks<Fr>), dim3((unsigned)((sizes[l] + 127) / 128)), dim3(128), 0, st->s, buf[l - 1],
                           (uint32_t)sizes[l - 1], buf[l]);
    for (uint32_t l = levels; l-- > 0;) {
        size_t chunks = (sizes[l] + GP_K - 1) / GP_K;
        hipLaunchKernelGGL((k_pp_apply<Fr>), dim3((unsigned)((chunks + 127) / 128)), dim3(128), 0, st->s, buf[l],
                           (uint32_t)sizes[l], (l + 1 < levels) ? buf[l + 1] : (const uint32_t*)nullptr);
    }
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

// ---- permutation product without per-chunk inversions ---------------------------------------------
// z[j+1] = prod_{k<=j} a_k / prod_{k<=j} b_k.  With PA = prefix products of the numerators and PR = prefix products of
// the REVERSED denominators, 1 / prod_{k<=j} b_k = (prod_{k>j} b_k) / D = PR[n-3-j] / D, D = PR[n-2] the product of all
// denominators: two scans, ONE inversion, two products per row.  (k_gp_perm_ratio inverts once per 8 rows: 24 field
// elements live per lane, 2 waves per SIMD, a divergent binary-Euclid loop -- 1.17 ms at n = 2^20; this path 0.4 ms.)
template <class Fr>
__global__ __launch_bounds__(256) void k_gp_perm_terms(uint32_t n, uint32_t nw, const uint32_t* __restrict__ wires,
                                                       const uint32_t* __restrict__ idp,
                                                       const uint32_t* __restrict__ sig,
                                                       const uint32_t* __restrict__ cst, uint32_t* __restrict__ num,
                                                       uint32_t* __restrict__ den_rev) {
    constexpr int N = Fr::N;
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j + 1 >= n) return;
    const Fp<Fr> beta = fp_load<Fr>(cst), gamma = fp_load<Fr>(cst + N);
    Fp<Fr> a = Fp<Fr>::one(), b = Fp<Fr>::one();
    for (uint32_t i = 0; i < nw; i++) {
        const size_t e = ((size_t)i * n + j) * N;
        Fp<Fr> t = fp_add<Fr>(fp_load<Fr>(wires + e), gamma);
        a = fp_mul<Fr>(a, fp_add<Fr>(t, fp_mul<Fr>(beta, fp_load<Fr>(idp + e))));
        b = fp_mul<Fr>(b, fp_add<Fr>(t, fp_mul<Fr>(beta, fp_load<Fr>(sig + e))));
    }
    fp_store<Fr>(num + (size_t)j * N, a);
    fp_store<Fr>(den_rev + (size_t)(n - 2 - j) * N, b);
}
template <class Fr>
__global__ void k_gp_inv_one(const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) fp_store<Fr>(out, fp_inv_safegcd<Fr>(fp_load<Fr>(in)));
}
template <class Fr>
__global__ __launch_bounds__(256) void k_gp_perm_combine(uint32_t n, const uint32_t* __restrict__ pa,
                                                         const uint32_t* __restrict__ pr,
                                                         const uint32_t* __restrict__ inv_total,
                                                         uint32_t* __restrict__ x) {
    constexpr int N = Fr::N;
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) fp_store<Fr>(x, Fp<Fr>::one());
    if (j + 1 >= n) return;
    Fp<Fr> r = fp_mul<Fr>(fp_load<Fr>(pa + (size_t)j * N), fp_load<Fr>(inv_total));
    if (j + 2 < n) r = fp_mul<Fr>(r, fp_load<Fr>(pr + (size_t)(n - 3 - j) * N));
    fp_store<Fr>(x + (size_t)(j + 1) * N, r);
}

template <class Fr>
static int32_t upload_beta_gamma(const uint64_t* beta, const uint64_t* gamma, uint32_t** d_cst, jfhip_stream* st) {
    JF_TRY(st->poly_c.ensure(2 * Fr::N * 4));
    uint64_t host[8];
    memcpy(host, beta, 32);
    memcpy(host + 4, gamma, 32);
    JF_HIP(hipMemcpyAsync(st->poly_c.p, host, 64, hipMemcpyHostToDevice, st->s));
    *d_cst = st->poly_c.as<uint32_t>();
    return JFHIP_OK;
}

template <class Fr>
static int32_t perm_product_t(uint32_t log_n, uint32_t nw, const uint32_t* wires, const uint32_t* idp,
                              const uint32_t* sig, const uint64_t* beta, const uint64_t* gamma, uint32_t* out,
                              jfhip_stream* st) {
    const size_t n = (size_t)1 << log_n;
    constexpr int N = Fr::N;
    uint32_t* d_cst;
    JF_TRY(upload_beta_gamma<Fr>(beta, gamma, &d_cst, st));
    static const bool per_chunk = getenv("JFHIP_GP_PER_CHUNK_INVERSION") != nullptr;   // the first formulation, kept for A/B runs
    if (per_chunk || n < 4) {
        const size_t lanes = (n + GP_INV_CHUNK - 1) / GP_INV_CHUNK;
        hipLaunchKernelGGL((k_gp_perm_ratio<Fr>), dim3((unsigned)((lanes + 127) / 128)), dim3(128), 0, st->s,
                           (uint32_t)n, nw, wires, idp, sig, d_cst, out);
        JF_HIP(hipGetLastError());
        return prefix_product_t<Fr>(out, n, st);
    }
    // numerators, reversed denominators (n - 1 each) and the inverse of the total, in the stream's scratch
    JF_TRY(st->poly_a.ensure((2 * n + 1) * N * 4));
    uint32_t* num = st->poly_a.as<uint32_t>();
    uint32_t* den_rev = num + n * N;
    uint32_t* inv_total = num + 2 * n * N;
    const dim3 grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL((k_gp_perm_terms<Fr>), grid, dim3(256), 0, st->s, (uint32_t)n, nw, wires, idp, sig, d_cst, num,
                       den_rev);
    JF_HIP(hipGetLastError());
    JF_TRY(prefix_product_t<Fr>(num, n - 1, st));
    JF_TRY(prefix_product_t<Fr>(den_rev, n - 1, st));
    hipLaunchKernelGGL((k_gp_inv_one<Fr>), dim3(1), dim3(64), 0, st->s, den_rev + (n - 2) * N, inv_total);
    hipLaunchKernelGGL((k_gp_perm_combine<Fr>), grid, dim3(256), 0, st->s, (uint32_t)n, num, den_rev, inv_total, out);
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

template <class Fr>
static int32_t lookup_product_t(uint32_t log_n, const uint32_t* f, const uint32_t* t, const uint32_t* s,
                                const uint64_t* beta, const uint64_t* gamma, uint32_t* out, jfhip_stream* st) {
    const size_t n = (size_t)1 << log_n;
    uint32_t* d_cst;
    JF_TRY(upload_beta_gamma<Fr>(beta, gamma, &d_cst, st));
    const size_t lanes = (n + GP_INV_CHUNK - 1) / GP_INV_CHUNK;
    hipLaunchKernelGGL((k_gp_lookup_ratio<Fr>), dim3((unsigned)((lanes + 127) / 128)), dim3(128), 0, st->s, (uint32_t)n,
                       f, t, s, d_cst, out);
    JF_HIP(hipGetLastError());
    if (n > 1) JF_TRY(prefix_product_t<Fr>(out, n - 1, st));
    hipLaunchKernelGGL((k_gp_set_one<Fr>), dim3(1), dim3(64), 0, st->s, out + (n - 1) * Fr::N);
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

// ---- Plookup merged vectors -----------------------------------------------------------------------
// merged_table_value / merged_lookup_wire_value (relation/src/constraint_system.rs:1648-1687) for every row of H:
//   table[i]  = range[i] + q_lookup[i] * tau * (table_dom_sep[i] + tau * (key[i] + tau * (w3[i] + tau * w4[i])))
//   lookup[i] = w5[i]    + q_lookup[i] * tau * (q_dom_sep[i]     + tau * (w0[i]  + tau * (w1[i] + tau * w2[i])))
// One lane per row, 11 loads and 2 stores of 32 B, 9 products: HBM-bound elementwise work.
template <class Fr>
__global__ __launch_bounds__(256) void k_lookup_merge(uint32_t n, const uint32_t* __restrict__ wires,
                                                      const uint32_t* __restrict__ tabs, FpArg tau_arg,
                                                      uint32_t* __restrict__ out_table,
                                                      uint32_t* __restrict__ out_lookup) {
    constexpr int N = Fr::N;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<Fr> tau;
    for (int k = 0; k < N; k++) tau.v[k] = tau_arg.v[k];
    auto W = [&](int w) { return fp_load<Fr>(wires + ((size_t)w * n + i) * N); };
    auto T = [&](int t) { return fp_load<Fr>(tabs + ((size_t)t * n + i) * N); };
    const Fp<Fr> ql_tau = fp_mul<Fr>(T(4), tau);
    Fp<Fr> a = fp_add<Fr>(W(3), fp_mul<Fr>(tau, W(4)));
    a = fp_add<Fr>(T(1), fp_mul<Fr>(tau, a));
    a = fp_add<Fr>(T(2), fp_mul<Fr>(tau, a));
    fp_store<Fr>(out_table + (size_t)i * N, fp_add<Fr>(T(0), fp_mul<Fr>(ql_tau, a)));
    Fp<Fr> b = fp_add<Fr>(W(1), fp_mul<Fr>(tau, W(2)));
    b = fp_add<Fr>(W(0), fp_mul<Fr>(tau, b));
    b = fp_add<Fr>(T(3), fp_mul<Fr>(tau, b));
    fp_store<Fr>(out_lookup + (size_t)i * N, fp_add<Fr>(W(5), fp_mul<Fr>(ql_tau, b)));
}

int32_t plonk_lookup_merge_device(int32_t curve, uint32_t log_n, const uint32_t* wires, const uint32_t* tabs,
                                  const uint64_t* tau, uint32_t* out_table, uint32_t* out_lookup, jfhip_stream* st) {
    if (log_n > 30) {
        set_error("lookup merge: bad size (log_n %u)", log_n);
        return JFHIP_ERR_INVALID;
    }
    const size_t n = (size_t)1 << log_n;
    FpArg t;
    memcpy(t.v, tau, 32);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (curve == JFHIP_BN254)
        hipLaunchKernelGGL((k_lookup_merge<Bn254Fr>), grid, dim3(256), 0, st->s, (uint32_t)n, wires, tabs, t, out_table,
                           out_lookup);
    else
        hipLaunchKernelGGL((k_lookup_merge<Bls381Fr>), grid, dim3(256), 0, st->s, (uint32_t)n, wires, tabs, t, out_table,
                           out_lookup);
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

int32_t plonk_perm_product_device(int32_t curve, uint32_t log_n, uint32_t nw, const uint32_t* wires,
                                  const uint32_t* idp, const uint32_t* sig, const uint64_t* beta, const uint64_t* gamma,
                                  uint32_t* out, jfhip_stream* st) {
    if (log_n > 30 || nw == 0 || nw > 8) {
        set_error("perm product: bad size (log_n %u, %u wire types)", log_n, nw);
        return JFHIP_ERR_INVALID;
    }
    if (curve == JFHIP_BN254) return perm_product_t<Bn254Fr>(log_n, nw, wires, idp, sig, beta, gamma, out, st);
    return perm_product_t<Bls381Fr>(log_n, nw, wires, idp, sig, beta, gamma, out, st);
}

int32_t plonk_lookup_product_device(int32_t curve, uint32_t log_n, const uint32_t* f, const uint32_t* t,
                                    const uint32_t* s, const uint64_t* beta, const uint64_t* gamma, uint32_t* out,
                                    jfhip_stream* st) {
    if (log_n == 0 || log_n > 30) {
        set_error("lookup product: bad size (log_n %u)", log_n);
        return JFHIP_ERR_INVALID;
    }
    if (curve == JFHIP_BN254) return lookup_product_t<Bn254Fr>(log_n, f, t, s, beta, gamma, out, st);
    return lookup_product_t<Bls381Fr>(log_n, f, t, s, beta, gamma, out, st);
}

}  // namespace jf
