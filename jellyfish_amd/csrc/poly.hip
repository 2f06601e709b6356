// Polynomial glue of UnivariateKzgPCS::open on the device (reference
// pcs/src/univariate_kzg/mod.rs:153-158,174): witness polynomial (f(X) - f(z)) / (X - z) by
// synthetic division and f(z) by Horner, both as one suffix recurrence
//     s_i = f_i + z * s_{i+1},  s_n = 0     =>  witness_{i-1} = s_i (i >= 1),  f(z) = s_0,
// evaluated as a chunked scan (chunk-local Horner values, recursive carry computation with
// multiplier z^K, then a second sweep), so the serial depth is O(K log_K n) instead of n.
// Also: element-wise field ops for the test surface and the peak-rate probes for bench.py.
#include "common.hpp"
#include "curve.hpp"
#include "fp29.hpp"

namespace jf {

constexpr uint32_t HORNER_K = 64;

// h[c] = sum_{k<K} in[cK + k] mu^k   (top chunk may be short)
template <class Fr>
__global__ __launch_bounds__(128) void k_horner_chunks(const uint32_t* __restrict__ in, uint32_t m,
                                                       const uint32_t* __restrict__ mu_p, uint32_t* __restrict__ h) {
    constexpr int N = Fr::N;
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t lo = (uint64_t)c * HORNER_K;
    if (lo >= m) return;
    uint32_t hi = (uint32_t)((lo + HORNER_K < m) ? lo + HORNER_K : m);
    Fp<Fr> mu = fp_load<Fr>(mu_p);
    Fp<Fr> acc = Fp<Fr>::zero();
    for (uint32_t k = hi; k-- > (uint32_t)lo;) acc = fp_add<Fr>(fp_mul<Fr>(acc, mu), fp_load<Fr>(in + (size_t)k * N));
    fp_store<Fr>(h + (size_t)c * N, acc);
}

// s[k] = in[k] + mu * s[k+1] inside chunk c, seeded with the suffix value entering the chunk:
// carry_src[c + 1] (the upper level's suffix array), or 0 for the top chunk / top level.
template <class Fr>
__global__ __launch_bounds__(128) void k_horner_apply(const uint32_t* __restrict__ in, uint32_t m,
                                                      const uint32_t* __restrict__ mu_p,
                                                      const uint32_t* __restrict__ carry_src, uint32_t n_carry,
                                                      uint32_t* __restrict__ s) {
    constexpr int N = Fr::N;
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t lo = (uint64_t)c * HORNER_K;
    if (lo >= m) return;
    uint32_t hi = (uint32_t)((lo + HORNER_K < m) ? lo + HORNER_K : m);
    Fp<Fr> mu = fp_load<Fr>(mu_p);
    Fp<Fr> acc = Fp<Fr>::zero();
    if (carry_src && c + 1 < n_carry) acc = fp_load<Fr>(carry_src + (size_t)(c + 1) * N);
    for (uint32_t k = hi; k-- > (uint32_t)lo;) {
        acc = fp_add<Fr>(fp_mul<Fr>(acc, mu), fp_load<Fr>(in + (size_t)k * N));
        fp_store<Fr>(s + (size_t)k * N, acc);
    }
}

// mus[l] = z^(K^l), l < levels
template <class Fr>
__global__ void k_horner_mus(const uint32_t* __restrict__ z, uint32_t levels, uint32_t* __restrict__ mus) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Fp<Fr> mu = fp_load<Fr>(z);
    for (uint32_t l = 0; l < levels; l++) {
        fp_store<Fr>(mus + (size_t)l * Fr::N, mu);
        mu = fp_pow_u64<Fr>(mu, HORNER_K);
    }
}

// By-value field element for kernels whose scalar argument must not pass through device memory first.
struct FrArg {
    uint32_t v[8];
};
template <class Fr>
__global__ void k_horner_mus_val(FrArg z, uint32_t levels, uint32_t* __restrict__ mus) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Fp<Fr> mu;
    for (int i = 0; i < Fr::N; i++) mu.v[i] = z.v[i];
    for (uint32_t l = 0; l < levels; l++) {
        fp_store<Fr>(mus + (size_t)l * Fr::N, mu);
        mu = fp_pow_u64<Fr>(mu, HORNER_K);
    }
}

constexpr uint32_t HORNER_MAX_LEVELS = 8;  // 64^8 > 2^32 coefficients

// level sizes of the chunked scan over n coefficients; returns the number of levels
static inline uint32_t horner_levels(size_t n, size_t* sizes) {
    uint32_t levels = 0;
    sizes[levels++] = n;
    while (sizes[levels - 1] > HORNER_K && levels < HORNER_MAX_LEVELS) {
        sizes[levels] = (sizes[levels - 1] + HORNER_K - 1) / HORNER_K;
        levels++;
    }
    return levels;
}
// 32-bit words of workspace the scan over n coefficients needs (chunk values + their suffix arrays of levels 1..,
// one multiplier per level)
static inline size_t horner_workspace_words(size_t n) {
    size_t sizes[HORNER_MAX_LEVELS];
    const uint32_t levels = horner_levels(n, sizes);
    size_t extra = 0;
    for (uint32_t l = 1; l < levels; l++) extra += sizes[l];
    return (extra * 2 + levels + 1) * 8;
}

// The scan itself on a caller-provided workspace `ws` (horner_workspace_words(n) words).  The point comes either from
// device memory (d_point) or by value (z_val, when d_point is NULL).  Allocates nothing, host or device.
template <class Fr>
static int32_t open_witness_ws(const uint32_t* d_coeffs, size_t n, const uint32_t* d_point, const FrArg* z_val,
                               uint32_t* ws, uint32_t* d_suffix, hipStream_t stream) {
    // d_suffix receives s_0 .. s_{n-1}; the witness is d_suffix + one element, eval is s_0
    constexpr int N = Fr::N;
    size_t sizes[HORNER_MAX_LEVELS];
    const uint32_t levels = horner_levels(n, sizes);
    size_t extra = 0;
    for (uint32_t l = 1; l < levels; l++) extra += sizes[l];
    uint32_t* hbuf = ws;                // chunk values of levels 1..
    uint32_t* sbuf = hbuf + extra * N;  // their suffix arrays
    uint32_t* mus = sbuf + extra * N;
    if (d_point)
        hipLaunchKernelGGL((k_horner_mus<Fr>), dim3(1), dim3(64), 0, stream, d_point, levels, mus);
    else
        hipLaunchKernelGGL((k_horner_mus_val<Fr>), dim3(1), dim3(64), 0, stream, *z_val, levels, mus);
    uint32_t *h[HORNER_MAX_LEVELS], *s[HORNER_MAX_LEVELS];
    h[0] = const_cast<uint32_t*>(d_coeffs);
    s[0] = d_suffix;
    size_t o = 0;
    for (uint32_t l = 1; l < levels; l++) {
        h[l] = hbuf + o * N;
        s[l] = sbuf + o * N;
        o += sizes[l];
    }
    for (uint32_t l = 1; l < levels; l++) {
        unsigned blocks = (unsigned)((sizes[l] + 127) / 128);
        hipLaunchKernelGGL((k_horner_chunks<Fr>), dim3(blocks), dim3(128), 0, stream, h[l - 1], (uint32_t)sizes[l - 1],
                           mus + (size_t)(l - 1) * N, h[l]);
    }
    for (uint32_t l = levels; l-- > 0;) {
        size_t chunks = (sizes[l] + HORNER_K - 1) / HORNER_K;
        unsigned blocks = (unsigned)((chunks + 127) / 128);
        const uint32_t* carry = (l + 1 < levels) ? s[l + 1] : nullptr;
        uint32_t n_carry = (l + 1 < levels) ? (uint32_t)sizes[l + 1] : 0u;
        hipLaunchKernelGGL((k_horner_apply<Fr>), dim3(blocks), dim3(128), 0, stream, h[l], (uint32_t)sizes[l],
                           mus + (size_t)l * N, carry, n_carry, s[l]);
    }
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

int32_t open_witness_device(int32_t curve, const uint32_t* d_coeffs, size_t n, const uint32_t* d_point,
                            uint32_t* d_witness, uint32_t* d_eval, jfhip_stream* st) {
    // d_witness: n elements of scratch-and-result: [0] = eval, [1..n) = witness coefficients
    (void)d_eval;
    if (n == 0) return JFHIP_OK;
    if (n >= ((size_t)1 << 32)) {
        set_error("divide: polynomial too long");
        return JFHIP_ERR_INVALID;
    }
    JF_TRY(st->poly_b.ensure(horner_workspace_words(n) * 4));
    uint32_t* ws = st->poly_b.as<uint32_t>();
    if (curve == JFHIP_BN254) return open_witness_ws<Bn254Fr>(d_coeffs, n, d_point, nullptr, ws, d_witness, st->s);
    return open_witness_ws<Bls381Fr>(d_coeffs, n, d_point, nullptr, ws, d_witness, st->s);
}

// ---- the entry point that allocates nothing (jfhip_poly_divide_linear_noalloc_device) ------------------------------
// For callers that cannot call hipMalloc (HIP graph capture, tight serving loops, harnesses that forbid it): the same
// chunked scan as above with its workspace (chunk values, suffix arrays, per-level multipliers) in a file-scope device
// array instead of the stream's grow-only DevBuf.  Every GPU holds its own copy of the array at its own address; the
// address is taken once per device with hipGetSymbolAddress and only that pointer is ever handed to kernels (the array's
// name in host code is a host-side stand-in, not a device address).
constexpr size_t NOALLOC_MAX_N = (size_t)1 << 23;
// levels of 2^23 coefficients: 131072 + 2048 + 32 chunk values, twice (values + suffixes), + 5 multipliers
constexpr size_t NOALLOC_SCRATCH_WORDS = ((131072 + 2048 + 32) * 2 + 5) * 8;  // 8.13 MiB
static __device__ __align__(256) uint32_t scratch[NOALLOC_SCRATCH_WORDS];

constexpr int NOALLOC_MAX_DEVICES = 64;
static std::mutex g_noalloc_mu;
static uint32_t* g_noalloc_ptr[NOALLOC_MAX_DEVICES] = {};
// Calls share their GPU's array.  Two calls on different streams of one GPU are ORDERED, not raced: a per-device host
// mutex serialises the enqueue, and the next call's stream waits on an event recorded after the previous call's last
// kernel (events are created at set-up; waiting and recording allocate nothing).  A stream under graph capture takes no
// part in that (an event recorded outside the capture cannot be waited on inside it): such callers serialise themselves.
static std::mutex g_noalloc_dev_mu[NOALLOC_MAX_DEVICES];
static hipEvent_t g_noalloc_done[NOALLOC_MAX_DEVICES] = {};
static bool g_noalloc_recorded[NOALLOC_MAX_DEVICES] = {};

// once per device: the device address of THAT GPU's copy of `scratch` (taken with the device current, restored after)
int32_t noalloc_setup_device(int dev, uint32_t** out) {
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) {
        set_error("no-alloc entry point: device ordinal %d beyond the %d supported", dev, NOALLOC_MAX_DEVICES);
        return JFHIP_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(g_noalloc_mu);
    if (!g_noalloc_ptr[dev]) {
        int cur = -1;
        JF_HIP(hipGetDevice(&cur));
        if (cur != dev) JF_HIP(hipSetDevice(dev));
        uint32_t* ptr = nullptr;
        hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        hipEvent_t ev = nullptr;
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (cur != dev) (void)hipSetDevice(cur);
        JF_HIP(e);
        if (!ptr) {
            set_error("hipGetSymbolAddress(scratch) returned a null address on device %d", dev);
            return JFHIP_ERR_DEVICE;
        }
        g_noalloc_done[dev] = ev;
        g_noalloc_ptr[dev] = ptr;
    }
    if (out) *out = g_noalloc_ptr[dev];
    return JFHIP_OK;
}

int32_t poly_divide_linear_noalloc(int32_t curve, const uint32_t* d_coeffs, size_t n, const uint64_t* point_mont,
                                   uint32_t* d_out, hipStream_t stream) {
    if (n > NOALLOC_MAX_N) {
        set_error("no-alloc divide: %zu coefficients exceed the fixed workspace (at most %zu); use "
                  "jfhip_poly_divide_linear_device", n, NOALLOC_MAX_N);
        return JFHIP_ERR_INVALID;
    }
    if (horner_workspace_words(n) > NOALLOC_SCRATCH_WORDS) {  // cannot happen for n <= NOALLOC_MAX_N; host-side guard
        set_error("no-alloc divide: workspace bound violated for n = %zu", n);
        return JFHIP_ERR_INVALID;
    }
    // the array of the GPU that owns `stream` (not of the calling thread's current device: a thread whose current
    // device is B may pass a stream of device A, and the kernels run where the stream lives)
    hipDevice_t sdev = 0;
    JF_HIP(hipStreamGetDevice(stream, &sdev));
    const int dev = (int)sdev;
    uint32_t* ptr = nullptr;
    JF_TRY(noalloc_setup_device(dev, &ptr));
    FrArg z;
    memcpy(z.v, point_mont, 32);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    if (!capturing) (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(g_noalloc_dev_mu[dev]);
    if (!capturing && g_noalloc_recorded[dev]) JF_HIP(hipStreamWaitEvent(stream, g_noalloc_done[dev], 0));
    // nothing in the workspace has to start at zero: every level writes its chunk values and suffixes before the next
    // kernel reads them, and the top chunk's carry is the constant 0, not a stored value
    int32_t rc;
    if (curve == JFHIP_BN254) rc = open_witness_ws<Bn254Fr>(d_coeffs, n, nullptr, &z, ptr, d_out, stream);
    else rc = open_witness_ws<Bls381Fr>(d_coeffs, n, nullptr, &z, ptr, d_out, stream);
    if (!capturing) {
        // recorded on every path: kernels enqueued before a launch error still use the array
        hipError_t e = hipEventRecord(g_noalloc_done[dev], stream);
        if (e == hipSuccess) g_noalloc_recorded[dev] = true;
        else if (rc == JFHIP_OK) JF_HIP(e);
    }
    return rc;
}

// ---- batched evaluation at one point --------------------------------------------------------------
// f_p(z) for up to 32 polynomials (prover.rs:205-229 evaluates ~10 polynomials at zeta): the chunk
// reduction of the scan above without the second sweep, all polynomials in one launch per level.
constexpr int EVAL_MAX = 32;
struct EvalArgs {
    const uint32_t* in[EVAL_MAX];
    uint32_t* out[EVAL_MAX];
    uint32_t len[EVAL_MAX];
};
template <class Fr>
__global__ __launch_bounds__(128) void k_horner_chunks_batch(EvalArgs a, const uint32_t* __restrict__ mu_p) {
    constexpr int N = Fr::N;
    const uint32_t* in = a.in[blockIdx.y];
    const uint32_t m = a.len[blockIdx.y];
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t lo = (uint64_t)c * HORNER_K;
    if (lo >= m) return;
    uint32_t hi = (uint32_t)((lo + HORNER_K < m) ? lo + HORNER_K : m);
    Fp<Fr> mu = fp_load<Fr>(mu_p);
    Fp<Fr> acc = Fp<Fr>::zero();
    for (uint32_t k = hi; k-- > (uint32_t)lo;) acc = fp_add<Fr>(fp_mul<Fr>(acc, mu), fp_load<Fr>(in + (size_t)k * N));
    fp_store<Fr>(a.out[blockIdx.y] + (size_t)c * N, acc);
}

template <class Fr>
static int32_t poly_eval_batch_t(const uint32_t* const* d_polys, const size_t* lens, size_t count,
                                 const uint32_t* d_point, uint32_t* d_out, jfhip_stream* st) {
    constexpr int N = Fr::N;
    size_t maxlen = 0;
    for (size_t p = 0; p < count; p++) maxlen = lens[p] > maxlen ? lens[p] : maxlen;
    if (maxlen >= ((size_t)1 << 32)) {
        set_error("eval: polynomial too long");
        return JFHIP_ERR_INVALID;
    }
    uint32_t levels = 1;
    for (size_t m = maxlen; m > 1; m = (m + HORNER_K - 1) / HORNER_K) levels++;
    const size_t per_poly = (maxlen + HORNER_K - 1) / HORNER_K + 1;  // level-1 size bounds all later levels
    JF_TRY(st->poly_b.ensure((2 * per_poly * count + levels + 1) * N * 4));
    uint32_t* buf0 = st->poly_b.as<uint32_t>();
    uint32_t* buf1 = buf0 + per_poly * count * N;
    uint32_t* mus = buf1 + per_poly * count * N;
    hipLaunchKernelGGL((k_horner_mus<Fr>), dim3(1), dim3(64), 0, st->s, d_point, levels, mus);
    EvalArgs a;
    std::vector<size_t> cur(count);
    for (size_t p = 0; p < count; p++) {
        a.in[p] = d_polys[p];
        cur[p] = lens[p];
    }
    JF_HIP(hipMemsetAsync(d_out, 0, count * N * 4, st->s));  // empty polynomials evaluate to 0
    uint32_t level = 0;
    for (;;) {
        size_t maxcur = 0;
        for (size_t p = 0; p < count; p++) maxcur = cur[p] > maxcur ? cur[p] : maxcur;
        if (maxcur == 0) break;
        const bool last = maxcur <= HORNER_K;
        uint32_t* dst = (level & 1) ? buf1 : buf0;
        for (size_t p = 0; p < count; p++) {
            a.len[p] = (uint32_t)cur[p];
            a.out[p] = last ? d_out + p * N : dst + p * per_poly * N;
        }
        const size_t chunks = (maxcur + HORNER_K - 1) / HORNER_K;
        hipLaunchKernelGGL((k_horner_chunks_batch<Fr>), dim3((unsigned)((chunks + 127) / 128), (unsigned)count),
                           dim3(128), 0, st->s, a, mus + (size_t)level * N);
        if (last) break;
        for (size_t p = 0; p < count; p++) {
            a.in[p] = a.out[p];
            cur[p] = (cur[p] + HORNER_K - 1) / HORNER_K;
        }
        level++;
    }
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

int32_t poly_eval_batch_device(int32_t curve, const uint32_t* const* d_polys, const size_t* lens, size_t count,
                               const uint32_t* d_point, uint32_t* d_out, jfhip_stream* st) {
    if (count == 0) return JFHIP_OK;
    if (count > (size_t)EVAL_MAX) {
        set_error("eval: at most %d polynomials per call", EVAL_MAX);
        return JFHIP_ERR_INVALID;
    }
    if (curve == JFHIP_BN254) return poly_eval_batch_t<Bn254Fr>(d_polys, lens, count, d_point, d_out, st);
    return poly_eval_batch_t<Bls381Fr>(d_polys, lens, count, d_point, d_out, st);
}

// ---- linear combination of polynomials ----------------------------------------------------------
// out[i] = sum_k c_k * p_k[i]: the prover's linearisation polynomial and batched opening polynomial
// (reference plonk/src/proof_system/prover.rs:299-352, 470-488, mul_poly :1090-1099) without the
// polynomials leaving the device.  Ragged lengths: a polynomial contributes zero beyond its length.
constexpr int LINCOMB_MAX = 32;
struct LincombArgs {
    const uint32_t* polys[LINCOMB_MAX];
    uint32_t lens[LINCOMB_MAX];
    uint32_t count;
};

template <class Fr>
__global__ __launch_bounds__(256) void k_poly_lincomb(LincombArgs a, const uint32_t* __restrict__ coeffs, uint32_t n,
                                                      uint32_t* __restrict__ out) {
    constexpr int N = Fr::N;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<Fr> acc = Fp<Fr>::zero();
    for (uint32_t k = 0; k < a.count; k++) {
        if (i < a.lens[k]) {
            Fp<Fr> v = fp_load<Fr>(a.polys[k] + (size_t)i * N);
            acc = fp_add<Fr>(acc, fp_mul<Fr>(v, fp_load<Fr>(coeffs + (size_t)k * N)));
        }
    }
    fp_store<Fr>(out + (size_t)i * N, acc);
}

int32_t poly_lincomb_device(int32_t curve, const uint32_t* const* d_polys, const size_t* lens, size_t count,
                            const uint64_t* coeffs_mont, uint32_t* d_out, size_t n, jfhip_stream* st) {
    if (count == 0 || count > (size_t)LINCOMB_MAX) {
        set_error("lincomb: between 1 and %d polynomials per call", LINCOMB_MAX);
        return JFHIP_ERR_INVALID;
    }
    if (n >= ((size_t)1 << 32)) {
        set_error("lincomb: polynomial too long");
        return JFHIP_ERR_INVALID;
    }
    LincombArgs a;
    a.count = (uint32_t)count;
    for (size_t k = 0; k < count; k++) {
        a.polys[k] = d_polys[k];
        a.lens[k] = (uint32_t)(lens[k] < n ? lens[k] : n);
    }
    JF_TRY(st->poly_b.ensure(count * 32));
    JF_HIP(hipMemcpyAsync(st->poly_b.p, coeffs_mont, count * 32, hipMemcpyHostToDevice, st->s));
    dim3 grid((unsigned)((n + 255) / 256));
    if (n == 0) return JFHIP_OK;
    if (curve == JFHIP_BN254)
        hipLaunchKernelGGL((k_poly_lincomb<Bn254Fr>), grid, dim3(256), 0, st->s, a, st->poly_b.as<uint32_t>(),
                           (uint32_t)n, d_out);
    else
        hipLaunchKernelGGL((k_poly_lincomb<Bls381Fr>), grid, dim3(256), 0, st->s, a, st->poly_b.as<uint32_t>(),
                           (uint32_t)n, d_out);
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

// ---- multilinear fold (MultilinearKzgPCS::open, pcs/src/multilinear_kzg/mod.rs:330-338) ------------
// One round of open_internal on the evaluation table: q[b] = f[2b + 1] - f[2b], r[b] = f[2b] + q[b] * p, b < half.
// Lane b reads the adjacent pair (64 B) and writes two elements: a streaming pass, 128 B and one product per lane.
template <class Fr>
__global__ __launch_bounds__(256) void k_mle_fold(const uint32_t* __restrict__ f, uint32_t half,
                                                  const uint32_t* __restrict__ point, uint32_t* __restrict__ q,
                                                  uint32_t* __restrict__ r) {
    constexpr int N = Fr::N;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= half) return;
    const Fp<Fr> f0 = fp_load<Fr>(f + (size_t)(2 * (size_t)b) * N);
    const Fp<Fr> f1 = fp_load<Fr>(f + (size_t)(2 * (size_t)b + 1) * N);
    const Fp<Fr> d = fp_sub<Fr>(f1, f0);
    fp_store<Fr>(q + (size_t)b * N, d);
    fp_store<Fr>(r + (size_t)b * N, fp_add<Fr>(f0, fp_mul<Fr>(d, fp_load<Fr>(point))));
}

int32_t mle_fold_device(int32_t curve, const uint32_t* d_f, size_t half, const uint32_t* d_point, uint32_t* d_q,
                        uint32_t* d_r, jfhip_stream* st) {
    if (half == 0) return JFHIP_OK;
    if (half >= ((size_t)1 << 31)) {
        set_error("mle fold: table too long");
        return JFHIP_ERR_INVALID;
    }
    dim3 grid((unsigned)((half + 255) / 256));
    if (curve == JFHIP_BN254)
        hipLaunchKernelGGL((k_mle_fold<Bn254Fr>), grid, dim3(256), 0, st->s, d_f, (uint32_t)half, d_point, d_q, d_r);
    else
        hipLaunchKernelGGL((k_mle_fold<Bls381Fr>), grid, dim3(256), 0, st->s, d_f, (uint32_t)half, d_point, d_q, d_r);
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

// ---- element-wise field ops (test surface) ------------------------------------------------------
template <class C>
__global__ __launch_bounds__(128) void k_field_op(int op, const uint32_t* __restrict__ a,
                                                  const uint32_t* __restrict__ b, uint32_t* __restrict__ out,
                                                  uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<C> x = fp_load<C>(a + (size_t)i * C::N);
    Fp<C> y = (op <= 2 || op >= 7) ? fp_load<C>(b + (size_t)i * C::N) : Fp<C>::zero();
    Fp<C> r;
    switch (op) {
    case 0: r = fp_add<C>(x, y); break;
    case 1: r = fp_sub<C>(x, y); break;
    case 2: r = fp_mul<C>(x, y); break;
    case 3: r = fp_neg<C>(x); break;
    case 4: r = fp_inv<C>(x); break;
    case 5: r = fp_from_mont<C>(x); break;
    case 12: r = fp_inv_safegcd<C>(x); break;   // the inversion of the normalisation / batch-inversion kernels
    case 13: r = fp_inv_binary<C>(x); break;
    // 29-bit-limb arithmetic of fp29.hpp, through the conversions (test surface)
    case 7: r = f29_to_fp<C>(f29_mul<C>(f29_from_fp<C>(x), f29_from_fp<C>(y))); break;
    case 8: r = f29_to_fp<C>(f29_sqr<C>(f29_from_fp<C>(x))); break;
    case 9: r = f29_to_fp<C>(f29_sub<C, 2>(f29_from_fp<C>(x), f29_from_fp<C>(y))); break;
    case 10: {  // a lazy chain: ((x y - 2 x) * (x^2 - y + 8p)) * y - x y
        F29<C> a = f29_from_fp<C>(x), b = f29_from_fp<C>(y);
        F29<C> ab = f29_mul<C>(a, b);
        F29<C> t = f29_sub<C, 4>(ab, f29_normalize<C>(f29_dbl_lazy<C>(a)));
        F29<C> u = f29_sub<C, 8>(f29_sqr<C>(a), b);
        F29<C> w = f29_mul<C>(f29_mul<C>(t, u), b);
        r = f29_to_fp<C>(f29_sub<C, 4>(w, ab));
        break;
    }
    case 11: {  // x y + (x^2 - y + 8p) y with one reduction
        F29<C> a = f29_from_fp<C>(x), b = f29_from_fp<C>(y);
        r = f29_to_fp<C>(f29_dot2<C>(a, b, f29_sub<C, 8>(f29_sqr<C>(a), b), b));
        break;
    }
    default: r = fp_to_mont<C>(x); break;
    }
    fp_store<C>(out + (size_t)i * C::N, r);
}

int32_t field_op_device(int32_t field_id, int32_t op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n,
                        hipStream_t s) {
    dim3 grid((unsigned)((n + 127) / 128)), blk(128);
    switch (field_id) {
    case 0: hipLaunchKernelGGL((k_field_op<Bn254Fr>), grid, blk, 0, s, op, a, b, out, (uint32_t)n); break;
    case 1: hipLaunchKernelGGL((k_field_op<Bn254Fq>), grid, blk, 0, s, op, a, b, out, (uint32_t)n); break;
    case 2: hipLaunchKernelGGL((k_field_op<Bls381Fr>), grid, blk, 0, s, op, a, b, out, (uint32_t)n); break;
    case 3: hipLaunchKernelGGL((k_field_op<Bls381Fq>), grid, blk, 0, s, op, a, b, out, (uint32_t)n); break;
    default: set_error("unknown field id %d", field_id); return JFHIP_ERR_INVALID;
    }
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

// ---- peak-rate probes ---------------------------------------------------------------------------
// 8 independent v_mad_u64_u32 chains per lane
__global__ __launch_bounds__(256) void k_probe_mad(uint32_t iters, uint32_t seed, uint64_t* __restrict__ out) {
    uint32_t a = seed + threadIdx.x * 2654435761u, b = (seed ^ blockIdx.x) | 1u;
    uint64_t c0 = a, c1 = b, c2 = a ^ b, c3 = a + b, c4 = a * 3u, c5 = b * 5u, c6 = a * 7u, c7 = b * 9u;
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            c0 = (uint64_t)(uint32_t)c0 * b + c0;
            c1 = (uint64_t)(uint32_t)c1 * a + c1;
            c2 = (uint64_t)(uint32_t)c2 * b + c2;
            c3 = (uint64_t)(uint32_t)c3 * a + c3;
            c4 = (uint64_t)(uint32_t)c4 * b + c4;
            c5 = (uint64_t)(uint32_t)c5 * a + c5;
            c6 = (uint64_t)(uint32_t)c6 * b + c6;
            c7 = (uint64_t)(uint32_t)c7 * a + c7;
        }
    }
    uint64_t r = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
    if (r == 0x1234567ull) out[0] = r;  // keep the chains live
}

__global__ __launch_bounds__(256) void k_probe_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

// random 64-byte gathers from a table of `npts` points, L per lane, at the bucket accumulation's occupancy (4 waves per
// SIMD): DEP = 1 issues a lane's next gather only after the previous one returned (one in flight per lane, like the
// accumulate loop without prefetch); DEP = 0 leaves them independent.
template <int DEP>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_probe_gather(
    const uint4* __restrict__ table, uint64_t npts, uint32_t L, uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint32_t h = t * 2654435761u + 12345u;
    for (uint32_t k = 0; k < L; k++) {
        h = h * 1664525u + 1013904223u + (DEP ? (acc.x & 1u) : 0u);
        const uint64_t idx = ((uint64_t)(h ^ (h >> 15)) * npts) >> 32;
        const uint4* p = table + idx * 4;
        uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc.x ^= a.x ^ b.y ^ c.z ^ d.w;
        acc.y += a.y + b.x;
        acc.z ^= c.x + d.y;
        acc.w += a.w ^ d.x;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) out[0] = acc.x;
}

template <class C>
__global__ __launch_bounds__(256) void k_probe_mul29(uint32_t iters, uint32_t seed, uint32_t* __restrict__ out) {
    F29<C> x = f29_from_fp<C>(Fp<C>::one()), y = f29_from_fp<C>(Fp<C>::r2());
    x.v[0] ^= (threadIdx.x & 1);
    y.v[1] ^= seed & 1;
    for (uint32_t i = 0; i < iters; i++) {
        x = f29_mul<C>(x, y);
        y = f29_mul<C>(y, x);
    }
    if (x.v[0] == 0x12345u && y.v[1] == 0x777u) out[0] = x.v[2];
}

template <class C>
__global__ __launch_bounds__(256) void k_probe_mul(uint32_t iters, uint32_t seed, uint32_t* __restrict__ out) {
    Fp<C> x = Fp<C>::one(), y = Fp<C>::r2();
    x.v[0] ^= (threadIdx.x & 1);
    y.v[1] ^= seed & 1;
    x = fp_reduce_once<C>(x);
    for (uint32_t i = 0; i < iters; i++) {
        x = fp_mul<C>(x, y);
        y = fp_mul<C>(y, x);
    }
    if (x.v[0] == 0x12345u && y.v[1] == 0x777u) out[0] = x.v[2];
}

int32_t probe_const_mul(int32_t kind, int32_t arg, double* out);   // probe_mfma.hip: kinds 6 and 7

int32_t probe(int32_t kind, int32_t arg, double* out) {
    if (kind == 6 || kind == 7) return probe_const_mul(kind, arg, out);
    hipEvent_t e0, e1;
    JF_HIP(hipEventCreate(&e0));
    JF_HIP(hipEventCreate(&e1));
    float ms = 0;
    double work = 0;
    void* scratch = nullptr;
    JF_HIP(hipMalloc(&scratch, 4096));
    if (kind == 0) {
        const uint32_t iters = 4096, blocks = 256 * 16;
        hipLaunchKernelGGL(k_probe_mad, dim3(blocks), dim3(256), 0, 0, 64u, 1u, (uint64_t*)scratch);
        JF_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_probe_mad, dim3(blocks), dim3(256), 0, 0, iters, 1u, (uint64_t*)scratch);
        JF_HIP(hipEventRecord(e1, 0));
        work = (double)blocks * 256.0 * iters * 32.0;
    } else if (kind == 1) {
        const size_t bytes = (size_t)1 << 30;
        void *a = nullptr, *b = nullptr;
        JF_HIP(hipMalloc(&a, bytes));
        JF_HIP(hipMalloc(&b, bytes));
        JF_HIP(hipMemset(a, 1, bytes));
        hipLaunchKernelGGL(k_probe_copy, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16);
        JF_HIP(hipEventRecord(e0, 0));
        for (int r = 0; r < 4; r++)
            hipLaunchKernelGGL(k_probe_copy, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16);
        JF_HIP(hipEventRecord(e1, 0));
        JF_HIP(hipEventSynchronize(e1));
        JF_HIP(hipFree(a));
        JF_HIP(hipFree(b));
        work = 4.0 * 2.0 * (double)bytes;
    } else if (kind == 2) {
        const uint32_t iters = 512, blocks = 256 * 8;
        auto launch = [&](uint32_t it) {
            switch (arg) {
            case 0: hipLaunchKernelGGL((k_probe_mul<Bn254Fr>), dim3(blocks), dim3(256), 0, 0, it, 1u, (uint32_t*)scratch); break;
            case 1: hipLaunchKernelGGL((k_probe_mul<Bn254Fq>), dim3(blocks), dim3(256), 0, 0, it, 1u, (uint32_t*)scratch); break;
            case 2: hipLaunchKernelGGL((k_probe_mul<Bls381Fr>), dim3(blocks), dim3(256), 0, 0, it, 1u, (uint32_t*)scratch); break;
            default: hipLaunchKernelGGL((k_probe_mul<Bls381Fq>), dim3(blocks), dim3(256), 0, 0, it, 1u, (uint32_t*)scratch); break;
            }
        };
        launch(8);
        JF_HIP(hipEventRecord(e0, 0));
        launch(iters);
        JF_HIP(hipEventRecord(e1, 0));
        work = (double)blocks * 256.0 * iters * 2.0;
    } else if (kind == 3) {
        const uint32_t iters = 512, blocks = 256 * 8;
        auto launch = [&](uint32_t it) {
            switch (arg) {
            case 0: hipLaunchKernelGGL((k_probe_mul29<Bn254Fr>), dim3(blocks), dim3(256), 0, 0, it, 1u, (uint32_t*)scratch); break;
            case 1: hipLaunchKernelGGL((k_probe_mul29<Bn254Fq>), dim3(blocks), dim3(256), 0, 0, it, 1u, (uint32_t*)scratch); break;
            case 2: hipLaunchKernelGGL((k_probe_mul29<Bls381Fr>), dim3(blocks), dim3(256), 0, 0, it, 1u, (uint32_t*)scratch); break;
            default: hipLaunchKernelGGL((k_probe_mul29<Bls381Fq>), dim3(blocks), dim3(256), 0, 0, it, 1u, (uint32_t*)scratch); break;
            }
        };
        launch(8);
        JF_HIP(hipEventRecord(e0, 0));
        launch(iters);
        JF_HIP(hipEventRecord(e1, 0));
        work = (double)blocks * 256.0 * iters * 2.0;
    } else if (kind == 4 || kind == 5) {
        // arg = table size in GiB; 2^24 x 12 gathers like the bench MSM
        const size_t bytes = (size_t)(arg > 0 ? arg : 1) << 30;
        void* tab = nullptr;
        JF_HIP(hipMalloc(&tab, bytes));
        JF_HIP(hipMemset(tab, 0x5a, bytes));
        const uint32_t L = 256, lanes = 786432;
        auto launch = [&]() {
            if (kind == 4) hipLaunchKernelGGL((k_probe_gather<1>), dim3(lanes / 128), dim3(128), 0, 0, (const uint4*)tab, (uint64_t)(bytes / 64), L, (uint32_t*)scratch);
            else hipLaunchKernelGGL((k_probe_gather<0>), dim3(lanes / 128), dim3(128), 0, 0, (const uint4*)tab, (uint64_t)(bytes / 64), L, (uint32_t*)scratch);
        };
        launch();
        JF_HIP(hipEventRecord(e0, 0));
        launch();
        JF_HIP(hipEventRecord(e1, 0));
        JF_HIP(hipEventSynchronize(e1));
        JF_HIP(hipFree(tab));
        work = (double)lanes * L;
    } else {
        set_error("unknown probe kind %d", kind);
        return JFHIP_ERR_INVALID;
    }
    JF_HIP(hipGetLastError());
    JF_HIP(hipEventSynchronize(e1));
    JF_HIP(hipEventElapsedTime(&ms, e0, e1));
    JF_HIP(hipFree(scratch));
    JF_HIP(hipEventDestroy(e0));
    JF_HIP(hipEventDestroy(e1));
    *out = work / ((double)ms * 1e-3);
    return JFHIP_OK;
}

// ---- witness gather (Arithmetization::compute_wire_polynomials, relation/src/constraint_system.rs:1422-1444, before its
// ifft): out[i] = witness[wire_variables[i]], 32-byte rows.  One lane per row, two 16-byte loads and stores; the table
// (num_vars x 32 B) is read at random and mostly stays in L2 / Infinity Cache, the output is written once, coalesced.
__global__ __launch_bounds__(256) void k_gather_rows32(const uint4* __restrict__ table, uint32_t rows,
                                                       const uint32_t* __restrict__ idx, size_t count,
                                                       uint4* __restrict__ out, uint32_t* __restrict__ bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t v = idx[i];
    if (v >= rows) {  // reported to the host instead of faulting
        *bad = 1u;
        v = 0;
    }
    const uint4 a = table[2 * (size_t)v], b = table[2 * (size_t)v + 1];
    out[2 * i] = a;
    out[2 * i + 1] = b;
}

int32_t witness_gather_device(const uint32_t* d_witness, size_t num_vars, const uint32_t* d_wire_variables, size_t count,
                              uint32_t* d_out, jfhip_stream* st) {
    if (count == 0) return JFHIP_OK;
    JF_TRY(st->poly_c.ensure(128));
    uint32_t* bad = st->poly_c.as<uint32_t>() + 24;  // past the 64 bytes of constants other calls keep at the front
    JF_HIP(hipMemsetAsync(bad, 0, 4, st->s));
    hipLaunchKernelGGL(k_gather_rows32, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st->s,
                       (const uint4*)d_witness, (uint32_t)num_vars, d_wire_variables, count, (uint4*)d_out, bad);
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

}  // namespace jf
