// Radix-2 (coset) NTT / iNTT over Fr for gfx950: the transforms jf-plonk's prover runs through
// ark-poly's Radix2EvaluationDomain (reference call sites plonk/src/proof_system/prover.rs:
// 534-573,658-660 and relation/src/constraint_system.rs:1369-1622).
//
// Semantics: natural order in and out,
//   forward  out[i] = sum_j in[j] (h w^i)^j,  inverse = exact inverse (n^-1, h^-j scaling),
//   w = TWO_ADIC_ROOT_OF_UNITY^(2^(s - log n)), input zero-extended from in_len.
//
// Structure: mixed-radix Cooley-Tukey over P = ceil(log n / 8) digit passes.  A pass loads a
// tile of 2048 elements (2^d-point transforms x 2048/2^d adjacent columns) into LDS with
// coalesced 32-byte-per-lane rows, runs d butterfly layers out of LDS, applies the inter-pass
// twiddle (two-level table, so no n-sized twiddle array is streamed from HBM) and stores.
// The last pass performs the digit-reversal as a transposed store whose inner run is the
// first digit, so stores stay in 2^d-element-aligned contiguous segments.  The coset
// pre-scale / inverse post-scale are fused into the first load / last store.
#include <stdlib.h>

#include <map>
#include <mutex>

#include "common.hpp"
#include "fp.hpp"
#include "fp29.hpp"

namespace jf {

constexpr int NTT_TILE_LOG = 11;  // 2048 elements x 32 B = 64 KiB of LDS
constexpr int NTT_TILE = 1 << NTT_TILE_LOG;
constexpr int NTT_MAX_DIGIT = 8;
constexpr int NTT_THREADS = 256;

struct NttTables {
    uint32_t* params = nullptr;    // [0] w_n  [1] w_256  [2] n^-1  [3] h (or h^-1)   (8 limbs each)
    uint32_t* tw_small = nullptr;  // w_256^k, k < 256
    uint32_t* tl = nullptr;        // w_n^i, i < 2^a
    uint32_t* th = nullptr;        // w_n^(i 2^a), i < 2^(log_n - a)
    uint32_t* cl = nullptr;        // h^i
    uint32_t* ch = nullptr;        // s0 * h^(i 2^a)
    uint32_t a = 0;
    int scale_mode = 0;  // 0 none, 1 constant n^-1, 2 two-level table
    size_t nbytes = 0;
    size_t bytes() const { return nbytes; }
    void release() {
        uint32_t** all[] = {&params, &tw_small, &tl, &th, &cl, &ch};
        for (uint32_t** q : all) {
            if (*q) (void)hipFree(*q);
            *q = nullptr;
        }
    }
};

struct NttPass {
    const uint32_t* src;
    uint32_t* dst;
    uint32_t log_n, outer, inner, dbits, log_M, in_len;  // batch = outer (flattened into the grid) x inner (looped)
    int first, final, inverse;
    uint32_t ndig;
    uint32_t dig[4];
    NttTables t;
};

JF_DEV uint32_t bitrev_n(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// lo[e mod 2^a] * hi[e >> a]; hi[0] is 1 unless the table carries a scale (ALWAYS_HI)
template <class Fr, bool ALWAYS_HI = false>
JF_DEV Fp<Fr> two_level(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi, uint32_t a, uint64_t e) {
    Fp<Fr> x = fp_load<Fr>(lo + (size_t)(e & ((1ull << a) - 1)) * Fr::N);
    uint64_t h = e >> a;
    if (ALWAYS_HI || h) x = fp_mul<Fr>(x, fp_load<Fr>(hi + (size_t)h * Fr::N));
    return x;
}

// One digit pass.  A workgroup owns one tile position (2^d-point transforms x C adjacent columns) and
// loops over `inner` polynomials of the batch at that position, so the per-element factors that do
// not depend on the polynomial -- inter-pass twiddle, coset pre-scale, inverse post-scale -- are
// built once (two-level table: 2 loads + 1 multiply) and reused from registers: one multiply per
// element and polynomial instead of two.
// Zero extension (first pass, in_len < n): only rows jp < 2^rb of the tile are non-zero; after the
// first s0 = d - rb butterfly layers every group of 2^s0 LDS rows would hold copies of one input
// (u + 0*w), so those layers are replaced by replicating the scaled input at load time.
template <class Fr>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass(NttPass P) {
    constexpr int N = Fr::N;
    constexpr int PER = NTT_TILE / NTT_THREADS;  // 8 elements per lane
    __shared__ uint32_t lds[NTT_TILE * N + 128 * N];
    uint32_t* tws = lds + NTT_TILE * N;
    const uint32_t tid = threadIdx.x;
    const uint32_t d = P.dbits;
    const uint32_t logC = NTT_TILE_LOG - d;
    const uint32_t C = 1u << logC;
    const uint64_t ncols = (uint64_t)P.outer << (P.log_n - d);
    const uint64_t g0 = (uint64_t)blockIdx.x << logC;
    const uint32_t log_S = P.log_M - d;           // stride of this digit inside its row
    const uint32_t log_rows = P.log_n - P.log_M;  // rows of length M per polynomial
    const uint64_t n = 1ull << P.log_n;
    const size_t poly_stride = (size_t)n * N;

    if (tid < 128) {
        Fp<Fr> w = fp_load<Fr>(P.t.tw_small + tid * N);
        fp_store<Fr>(tws + tid * N, w);
    }

    // column -> (poly, base element index inside the poly) for non-final passes
    auto col_base = [&](uint64_t g, uint32_t& poly, uint64_t& base, uint32_t& jrest) {
        jrest = (uint32_t)(g & ((1ull << log_S) - 1));
        uint64_t row = (g >> log_S) & ((1ull << log_rows) - 1);
        poly = (uint32_t)(g >> (log_S + log_rows));
        base = (row << P.log_M) + jrest;
    };
    // final pass: columns are rows of 2^d contiguous elements, enumerated with the first digit fastest
    const uint32_t R1bits = (P.ndig >= 2) ? P.dig[0] : 0;
    const uint32_t log_NR = P.log_n - d;  // rows per polynomial
    auto final_col = [&](uint64_t g, uint32_t& poly, uint64_t& row, uint64_t& outbase) {
        uint32_t k1 = (uint32_t)(g & ((1ull << R1bits) - 1));
        uint64_t rest = (g >> R1bits) & ((1ull << (log_NR - R1bits)) - 1);
        poly = (uint32_t)(g >> log_NR);
        row = ((uint64_t)k1 << (log_NR - R1bits)) + rest;
        // digit-reverse `rest` (digits 2..ndig-1, the earliest digit outermost)
        uint64_t rev = 0;
        uint32_t shift = 0, consumed = 0;
        uint32_t rest_bits = log_NR - R1bits;
        for (uint32_t q = 1; q + 1 < P.ndig; q++) {
            uint32_t bq = P.dig[q];
            consumed += bq;
            uint64_t kq = (rest >> (rest_bits - consumed)) & ((1ull << bq) - 1);
            rev |= kq << shift;
            shift += bq;
        }
        outbase = k1 + (rev << R1bits);
    };

    // ---- zero-extension plan: rows jp < 2^rb carry data, s0 layers are replaced by replication ----
    uint32_t rb = d;
    if (P.first && P.in_len < n) {
        // first pass: idx = (jp << log_S) + jrest with jrest >= jrest0 = first column of the tile
        uint64_t jrest0 = P.final ? 0 : (g0 & ((1ull << log_S) - 1));
        uint64_t nz = (P.in_len > jrest0) ? (((uint64_t)P.in_len - 1 - jrest0) >> log_S) + 1 : 1;
        rb = 0;
        while ((1ull << rb) < nz) rb++;
        if (rb > d) rb = d;
    }
    const uint32_t s0 = d - rb;
    const uint32_t n_src = C << rb;  // distinct source elements of the tile

    // ---- per-lane descriptors, independent of the polynomial -----------------------------------
    uint32_t ld_off[PER];   // element offset of the source (relative to the polynomial group)
    uint32_t ld_slot[PER];  // first LDS row << logC | col
    bool ld_ok[PER], ld_scale[PER];
    Fp<Fr> cf[PER];
    uint32_t st_off[PER];
    bool st_ok[PER], st_mul[PER];
    Fp<Fr> tw[PER];
#pragma unroll
    for (int it = 0; it < PER; it++) {
        uint32_t e = it * NTT_THREADS + tid;
        ld_ok[it] = false;
        ld_scale[it] = false;
        ld_off[it] = 0;
        ld_slot[it] = 0;
        if (e < n_src) {
            uint32_t col = e & (C - 1), jp = e >> logC;
            uint64_t g = g0 + col;
            ld_slot[it] = ((bitrev_n(jp, rb) << s0) << logC) | col;
            if (g < ncols) {
                uint32_t poly;
                uint64_t idx;
                if (!P.final || P.first) {
                    uint64_t base;
                    uint32_t jrest;
                    if (P.final) {
                        poly = (uint32_t)g;
                        base = 0;
                    } else col_base(g, poly, base, jrest);
                    idx = base + ((uint64_t)jp << log_S);
                } else {
                    uint64_t row, ob;
                    final_col(g, poly, row, ob);
                    idx = (row << d) + jp;
                }
                if (!P.first || idx < P.in_len) {
                    ld_ok[it] = true;
                    ld_off[it] = (uint32_t)((uint64_t)poly * n + idx);
                    if (P.first && !P.inverse && P.t.scale_mode == 2 && idx) {
                        ld_scale[it] = true;
                        cf[it] = two_level<Fr>(P.t.cl, P.t.ch, P.t.a, idx);
                    }
                }
            }
        }
        // store side
        uint32_t col = e & (C - 1), k = e >> logC;
        uint64_t g = g0 + col;
        st_ok[it] = g < ncols;
        st_mul[it] = false;
        st_off[it] = 0;
        if (st_ok[it]) {
            uint32_t poly;
            uint64_t pos;
            if (!P.final) {
                uint64_t base;
                uint32_t jrest;
                col_base(g, poly, base, jrest);
                pos = base + ((uint64_t)k << log_S);
                uint64_t E = ((uint64_t)k * jrest) << log_rows;  // exponent of w_n
                if (E) {
                    st_mul[it] = true;
                    tw[it] = two_level<Fr>(P.t.tl, P.t.th, P.t.a, E);
                }
            } else {
                if (P.ndig == 1) {
                    poly = (uint32_t)g;
                    pos = k;
                } else {
                    uint64_t row, ob;
                    final_col(g, poly, row, ob);
                    pos = ob + ((uint64_t)k << log_NR);
                }
                if (P.inverse && P.t.scale_mode == 1) {
                    st_mul[it] = true;
                    tw[it] = fp_load<Fr>(P.t.params + 2 * N);
                } else if (P.inverse && P.t.scale_mode == 2) {
                    st_mul[it] = true;
                    tw[it] = two_level<Fr, true>(P.t.cl, P.t.ch, P.t.a, pos);
                }
            }
            st_off[it] = (uint32_t)((uint64_t)poly * n + pos);
        }
    }

    for (uint32_t b = 0; b < P.inner; b++) {
        const uint32_t* src = P.src + (size_t)b * poly_stride;
        uint32_t* dst = P.dst + (size_t)b * poly_stride;
        // ---- load (bit-reversed along the transform axis, replicated over skipped layers) ------
#pragma unroll
        for (int it = 0; it < PER; it++) {
            uint32_t e = it * NTT_THREADS + tid;
            if (e < n_src) {
                Fp<Fr> x = Fp<Fr>::zero();
                if (ld_ok[it]) {
                    x = fp_load<Fr>(src + (size_t)ld_off[it] * N);
                    if (ld_scale[it]) x = fp_mul<Fr>(x, cf[it]);
                }
                uint32_t col = ld_slot[it] & (C - 1), r0 = ld_slot[it] >> logC;
                for (uint32_t t = 0; t < (1u << s0); t++) fp_store<Fr>(lds + (((size_t)(r0 + t) << logC) + col) * N, x);
            }
        }
        __syncthreads();

        // ---- butterfly layers s0+1 .. d (decimation in time) ------------------------------------
        for (uint32_t s = s0 + 1; s <= d; s++) {
            const uint32_t log_half = s - 1, half = 1u << log_half;
#pragma unroll 2
            for (int it = 0; it < NTT_TILE / 2 / NTT_THREADS; it++) {
                uint32_t bf = it * NTT_THREADS + tid;
                uint32_t col = bf & (C - 1), q = bf >> logC;
                uint32_t j = q & (half - 1), grp = q >> log_half;
                uint32_t i0 = (grp << s) + j, i1 = i0 + half;
                uint32_t* p0 = lds + (((size_t)i0 << logC) + col) * N;
                uint32_t* p1 = lds + (((size_t)i1 << logC) + col) * N;
                Fp<Fr> u = fp_load<Fr>(p0), v = fp_load<Fr>(p1);
                if (j) v = fp_mul<Fr>(v, fp_load<Fr>(tws + (size_t)(j << (7 - log_half)) * N));
                fp_store<Fr>(p0, fp_add<Fr>(u, v));
                fp_store<Fr>(p1, fp_sub<Fr>(u, v));
            }
            __syncthreads();
        }

        // ---- store ------------------------------------------------------------------------------
#pragma unroll
        for (int it = 0; it < PER; it++) {
            uint32_t e = it * NTT_THREADS + tid;
            uint32_t col = e & (C - 1), k = e >> logC;
            if (st_ok[it]) {
                Fp<Fr> x = fp_load<Fr>(lds + (((size_t)k << logC) + col) * N);
                if (st_mul[it]) x = fp_mul<Fr>(x, tw[it]);
                fp_store<Fr>(dst + (size_t)st_off[it] * N, x);
            }
        }
        __syncthreads();
    }
}

// ---- register-radix digit pass (digits of 5..8 bits, at least two passes) ----------------------
//
// Same tile, same global addressing and the same result as k_ntt_pass, but the 2^d-point column
// transforms run as three Cooley-Tukey stages of radix <= 8 held in registers (8 elements per lane),
// exchanged through LDS twice per tile instead of once per butterfly layer:
//   j = ja + RA*jb + RA*RB*jc  ->  k = kc + 8*kb + 8*RB*ka
//   stage 1: DFT_8 over jc                       (lane = (ja, jb, col))
//   stage 2: * w_D^(RA*jb*kc), DFT_RB over jb    (lane = (ja, kc, col))
//   stage 3: * w_D^(ja*(kc+8*kb)), DFT_RA over ja (lane = (kb, kc, col))
// A digit shorter than 3 bits fills its lane's 8 registers with adjacent column groups.  Every
// factor is a power of w_256 (table in LDS).  Exchanges use two 16-byte planes with a row pitch
// of 257 entries, so readers (consecutive lanes -> consecutive entries) and writers are
// bank-conflict free.  Zero extension (first pass, only the jc = 0 rows non-zero, the PLONK 8n
// coset shape): stage 1 degenerates to a broadcast, one load per lane.
template <class Fr>
JF_DEV void ntt_bfly(Fp<Fr>& a, Fp<Fr>& b) {
    Fp<Fr> t = fp_add<Fr>(a, b);
    b = fp_sub<Fr>(a, b);
    a = t;
}
// opaque copy of an index: keeps loop-invariant twiddle loads inside the per-polynomial loop (hoisted
// out of it they would pin ~130 registers and spill)
JF_DEV uint32_t launder(uint32_t x) {
    asm volatile("" : "+v"(x));
    return x;
}
template <class Fr>
JF_DEV Fp<Fr> lds_get(const uint4* pl, const uint4* ph, uint32_t idx) {
    Fp<Fr> x;
    uint4 a = pl[idx], b = ph[idx];
    x.v[0] = a.x, x.v[1] = a.y, x.v[2] = a.z, x.v[3] = a.w;
    x.v[4] = b.x, x.v[5] = b.y, x.v[6] = b.z, x.v[7] = b.w;
    return x;
}
template <class Fr>
JF_DEV void lds_put(uint4* pl, uint4* ph, uint32_t idx, const Fp<Fr>& x) {
    pl[idx] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    ph[idx] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}

// the w_256 table in LDS, entries in the 29-bit constant form of fp_mul_const29 (9 words: two 16-byte planes + one word)
struct TwLds {
    uint4* l;
    uint4* h;
    uint32_t* t;
};
template <class Fr>
JF_DEV F29<Fr> tw29_get(const TwLds& T, uint32_t idx) {
    static_assert(F29<Fr>::L == 9, "9-limb scalars");
    F29<Fr> x;
    const uint4 a = T.l[idx], b = T.h[idx];
    x.v[0] = a.x, x.v[1] = a.y, x.v[2] = a.z, x.v[3] = a.w;
    x.v[4] = b.x, x.v[5] = b.y, x.v[6] = b.z, x.v[7] = b.w;
    x.v[8] = T.t[idx];
    return x;
}
template <class Fr>
JF_DEV void tw29_put(const TwLds& T, uint32_t idx, const F29<Fr>& x) {
    T.l[idx] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    T.h[idx] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    T.t[idx] = x.v[8];
}

// natural-order DFT of size 4: Y[k] = sum_j b_j w4^(jk); constants come from the w_256 table in LDS
// right where they are used (short live ranges)
template <class Fr>
JF_DEV void ntt_dft4(Fp<Fr>& b0, Fp<Fr>& b1, Fp<Fr>& b2, Fp<Fr>& b3, const TwLds& T) {
    ntt_bfly<Fr>(b0, b2);
    ntt_bfly<Fr>(b1, b3);
    b3 = fp_mul_const29<Fr>(b3, tw29_get<Fr>(T, launder(64)));
    ntt_bfly<Fr>(b0, b1);
    ntt_bfly<Fr>(b2, b3);
    Fp<Fr> t = b1;  // (Y0, Y2, Y1, Y3) -> natural order
    b1 = b2;
    b2 = t;
}
template <class Fr>
JF_DEV void ntt_dft8(Fp<Fr> (&a)[8], const TwLds& T) {
    ntt_dft4<Fr>(a[0], a[2], a[4], a[6], T);
    ntt_dft4<Fr>(a[1], a[3], a[5], a[7], T);
    a[3] = fp_mul_const29<Fr>(a[3], tw29_get<Fr>(T, launder(32)));
    a[5] = fp_mul_const29<Fr>(a[5], tw29_get<Fr>(T, launder(64)));
    a[7] = fp_mul_const29<Fr>(a[7], tw29_get<Fr>(T, launder(96)));
    ntt_bfly<Fr>(a[0], a[1]);
    ntt_bfly<Fr>(a[2], a[3]);
    ntt_bfly<Fr>(a[4], a[5]);
    ntt_bfly<Fr>(a[6], a[7]);
    // X0..X3 = a0,a2,a4,a6 ; X4..X7 = a1,a3,a5,a7
    Fp<Fr> t1 = a[1], t2 = a[2], t3 = a[3], t4 = a[4], t5 = a[5], t6 = a[6];
    a[1] = t2;
    a[2] = t4;
    a[3] = t6;
    a[4] = t1;
    a[5] = t3;
    a[6] = t5;
}
// 2^(3-X) independent natural-order DFTs of size 2^X on registers r = digit | group << X
template <class Fr, int X>
JF_DEV void ntt_dft_regs(Fp<Fr> (&v)[8], const TwLds& T);

template <class Fr, int X>
JF_DEV void ntt_dft_regs(Fp<Fr> (&v)[8], const TwLds& T) {
    static_assert(Fr::N == 8, "register-radix NTT assumes 8-limb scalars");
    if constexpr (X == 3) {
        ntt_dft8<Fr>(v, T);
    } else if constexpr (X == 2) {
        ntt_dft4<Fr>(v[0], v[1], v[2], v[3], T);
        ntt_dft4<Fr>(v[4], v[5], v[6], v[7], T);
    } else if constexpr (X == 1) {
        ntt_bfly<Fr>(v[0], v[1]);
        ntt_bfly<Fr>(v[2], v[3]);
        ntt_bfly<Fr>(v[4], v[5]);
        ntt_bfly<Fr>(v[6], v[7]);
    }
}

template <class Fr, int DB>
__global__ __launch_bounds__(NTT_THREADS, 2) void k_ntt_pass_r8(NttPass P) {
    constexpr int N = Fr::N;
    constexpr int XB = (DB - 3 >= 3) ? 3 : DB - 3, XA = DB - 3 - XB;
    constexpr int RA = 1 << XA, RB = 1 << XB;
    constexpr int LOGC = NTT_TILE_LOG - DB, C = 1 << LOGC;
    constexpr int F2 = 3 - XB, F3 = 3 - XA;            // register bits not used by the digit: column groups
    constexpr int LC2 = LOGC - F2, LC3 = LOGC - F3;    // column bits carried by the lane index
    constexpr int PITCH = 257;
    static_assert(DB >= 5 && DB <= 8 && XA <= 2 && LC2 >= 0 && (XA == 0 || LC3 >= 0), "digit plan");
    __shared__ uint4 xl[8 * PITCH], xh[8 * PITCH], twl[256], twh[256];
    __shared__ uint32_t twt[256];
    const TwLds T{twl, twh, twt};

    const uint32_t tid = threadIdx.x;
    // every twiddle of the pass is a constant: kept in the 29-bit form, products by fp_mul_const29
    tw29_put<Fr>(T, tid, f29_const_from_fp<Fr>(fp_load<Fr>(P.t.tw_small + tid * N)));

    const uint32_t d = DB;
    const uint64_t ncols = (uint64_t)P.outer << (P.log_n - d);
    const uint64_t g0 = (uint64_t)blockIdx.x << LOGC;
    const uint32_t log_S = P.log_M - d;
    const uint32_t log_rows = P.log_n - P.log_M;
    const uint64_t n = 1ull << P.log_n;
    const size_t poly_stride = (size_t)n * N;
    const bool jfast = P.final != 0;  // final pass: rows are contiguous, lanes run along j on the load side

    auto col_base = [&](uint64_t g, uint32_t& poly, uint64_t& base, uint32_t& jrest) {
        jrest = (uint32_t)(g & ((1ull << log_S) - 1));
        uint64_t row = (g >> log_S) & ((1ull << log_rows) - 1);
        poly = (uint32_t)(g >> (log_S + log_rows));
        base = (row << P.log_M) + jrest;
    };
    const uint32_t R1bits = P.dig[0];
    const uint32_t log_NR = P.log_n - d;
    auto final_col = [&](uint64_t g, uint32_t& poly, uint64_t& row, uint64_t& outbase) {
        uint32_t k1 = (uint32_t)(g & ((1ull << R1bits) - 1));
        uint64_t rest = (g >> R1bits) & ((1ull << (log_NR - R1bits)) - 1);
        poly = (uint32_t)(g >> log_NR);
        row = ((uint64_t)k1 << (log_NR - R1bits)) + rest;
        uint64_t rev = 0;
        uint32_t shift = 0, consumed = 0;
        uint32_t rest_bits = log_NR - R1bits;
        for (uint32_t q = 1; q + 1 < P.ndig; q++) {
            uint32_t bq = P.dig[q];
            consumed += bq;
            uint64_t kq = (rest >> (rest_bits - consumed)) & ((1ull << bq) - 1);
            rev |= kq << shift;
            shift += bq;
        }
        outbase = k1 + (rev << R1bits);
    };

    // ---- lane coordinates in the three stages ----------------------------------------------------
    uint32_t col1, ja1, jb1;
    if (!jfast) {
        col1 = tid & (C - 1);
        ja1 = (tid >> LOGC) & (RA - 1);
        jb1 = tid >> (LOGC + XA);
    } else {
        ja1 = tid & (RA - 1);
        jb1 = (tid >> XA) & (RB - 1);
        col1 = tid >> (XA + XB);
    }
    // stage 2 keeps the column-fastest lane order when it is also the storing stage (XA == 0)
    const bool jfast2 = jfast && XA > 0;
    uint32_t coll2, kc2, ja2;
    if (!jfast2) {
        coll2 = tid & ((1u << LC2) - 1);
        kc2 = (tid >> LC2) & 7;
        ja2 = tid >> (LC2 + 3);
    } else {
        ja2 = tid & (RA - 1);
        kc2 = (tid >> XA) & 7;
        coll2 = tid >> (XA + 3);
    }
    const uint32_t coll3 = tid & ((1u << (XA ? LC3 : 0)) - 1);
    const uint32_t kc3 = (tid >> (XA ? LC3 : 0)) & 7;
    const uint32_t kb3 = tid >> ((XA ? LC3 : 0) + 3);

    // ---- zero-extension: only rows jc = 0 carry data -> stage 1 is a broadcast -------------------
    bool zext = false;
    if (P.first && P.in_len < n && (uint64_t)C <= (1ull << log_S)) {
        uint64_t jrest0 = g0 & ((1ull << log_S) - 1);
        uint64_t nz = (P.in_len > jrest0) ? (((uint64_t)P.in_len - 1 - jrest0) >> log_S) + 1 : 1;
        zext = nz <= (uint64_t)(RA * RB);
    }
    const bool has_cf = P.first && !P.inverse && P.t.scale_mode == 2;

    // ---- load descriptors (stage-1 lane: 8 rows jc of one column) ---------------------------------
    uint32_t ld_off[8];
    uint32_t ld_idx0 = 0, ld_step = 0;  // element index inside the polynomial of register 0, step per register
    uint32_t ld_mask = 0;
    F29<Fr> cf0;
#pragma unroll
    for (int i = 0; i < F29<Fr>::L; i++) cf0.v[i] = 0;
    {
        const uint64_t g = g0 + col1;
        const uint32_t j1 = ja1 + (jb1 << XA);
        uint32_t poly = 0;
        uint64_t idx0 = 0, step = 0;
        if (g < ncols) {
            if (!P.final) {
                uint64_t base;
                uint32_t jrest;
                col_base(g, poly, base, jrest);
                idx0 = base + ((uint64_t)j1 << log_S);
                step = 1ull << (XA + XB + log_S);
            } else {
                uint64_t row, ob;
                final_col(g, poly, row, ob);
                idx0 = (row << d) + j1;
                step = 1ull << (XA + XB);
            }
        }
        ld_idx0 = (uint32_t)idx0;
        ld_step = (uint32_t)step;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint64_t idx = idx0 + (uint64_t)r * step;
            bool ok = (g < ncols) && (!P.first || idx < P.in_len);
            ld_off[r] = ok ? (uint32_t)((uint64_t)poly * n + idx) : 0u;
            ld_mask |= ok ? (1u << r) : 0u;
        }
        if (has_cf && zext && (ld_mask & 1u)) cf0 = f29_const_from_fp<Fr>(two_level<Fr>(P.t.cl, P.t.ch, P.t.a, idx0));
    }

    // ---- store descriptors (last stage's lane layout) ---------------------------------------------
    uint32_t st_off[8];
    uint32_t st_mask = 0, tw_mask = 0;
    F29<Fr> tw[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint32_t k, col;
        if constexpr (XA > 0) {
            col = ((uint32_t)(r >> XA) << LC3) | coll3;
            k = kc3 + (kb3 << 3) + ((uint32_t)(r & (RA - 1)) << (3 + XB));
        } else {
            col = ((uint32_t)(r >> XB) << LC2) | coll2;
            k = kc2 + ((uint32_t)(r & (RB - 1)) << 3);
        }
        const uint64_t g = g0 + col;
        st_off[r] = 0;
#pragma unroll
        for (int i = 0; i < F29<Fr>::L; i++) tw[r].v[i] = 0;
        if (g < ncols) {
            st_mask |= 1u << r;
            uint32_t poly;
            uint64_t pos;
            if (!P.final) {
                uint64_t base;
                uint32_t jrest;
                col_base(g, poly, base, jrest);
                pos = base + ((uint64_t)k << log_S);
                uint64_t E = ((uint64_t)k * jrest) << log_rows;
                if (E) {
                    tw_mask |= 1u << r;
                    tw[r] = f29_const_from_fp<Fr>(two_level<Fr>(P.t.tl, P.t.th, P.t.a, E));
                }
            } else {
                uint64_t row, ob;
                final_col(g, poly, row, ob);
                pos = ob + ((uint64_t)k << log_NR);
                if (P.inverse && P.t.scale_mode == 1) {
                    tw_mask |= 1u << r;
                    tw[r] = f29_const_from_fp<Fr>(fp_load<Fr>(P.t.params + 2 * N));
                } else if (P.inverse && P.t.scale_mode == 2) {
                    tw_mask |= 1u << r;
                    tw[r] = f29_const_from_fp<Fr>(two_level<Fr, true>(P.t.cl, P.t.ch, P.t.a, pos));
                }
            }
            st_off[r] = (uint32_t)((uint64_t)poly * n + pos);
        }
    }
    __syncthreads();  // w_256 table visible

    for (uint32_t b = 0; b < P.inner; b++) {
        const uint32_t* src = P.src + (size_t)b * poly_stride;
        uint32_t* dst = P.dst + (size_t)b * poly_stride;
        Fp<Fr> v[8];
        if (!zext) {
            // ---- stage 1: DFT_8 over jc ------------------------------------------------------------
#pragma unroll
            for (int r = 0; r < 8; r++) {
                v[r] = Fp<Fr>::zero();
                if ((ld_mask >> r) & 1u) {
                    v[r] = fp_load<Fr>(src + (size_t)ld_off[r] * N);
                    if (has_cf) v[r] = fp_mul<Fr>(v[r], two_level<Fr>(P.t.cl, P.t.ch, P.t.a, ld_idx0 + (uint32_t)r * ld_step));
                }
            }
            ntt_dft_regs<Fr, 3>(v, T);
            {
                const uint32_t fill2 = col1 >> LC2, cw = col1 & ((1u << LC2) - 1);
                const uint32_t r2 = jb1 | (fill2 << XB);
                const uint32_t t2 = jfast2 ? (ja1 | (cw << (XA + 3))) : (cw | (ja1 << (LC2 + 3)));
                const uint32_t sh = jfast2 ? XA : LC2;
#pragma unroll
                for (int r = 0; r < 8; r++) lds_put<Fr>(xl, xh, r2 * PITCH + t2 + ((uint32_t)r << sh), v[r]);
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = lds_get<Fr>(xl, xh, r * PITCH + tid);
        } else {
            Fp<Fr> x = Fp<Fr>::zero();
            if (ld_mask & 1u) {
                x = fp_load<Fr>(src + (size_t)ld_off[0] * N);
                if (has_cf) x = fp_mul_const29<Fr>(x, cf0);
            }
            lds_put<Fr>(xl, xh, tid, x);  // first pass is never final: entry = col | ja << LOGC | jb << (LOGC + XA)
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t col = ((uint32_t)(r >> XB) << LC2) | coll2;
                v[r] = lds_get<Fr>(xl, xh, col | (ja2 << LOGC) | ((uint32_t)(r & (RB - 1)) << (LOGC + XA)));
            }
        }
        // ---- stage 2: * w_D^(RA jb kc), DFT_RB over jb ----------------------------------------------
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t jb = r & (RB - 1);
            if (jb) v[r] = fp_mul_const29<Fr>(v[r], tw29_get<Fr>(T, launder((jb * kc2) << (XA + 8 - DB))));
        }
        ntt_dft_regs<Fr, XB>(v, T);
        if constexpr (XA > 0) {
            __syncthreads();  // exchange-1 reads done
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t kb = r & (RB - 1);
                const uint32_t col = ((uint32_t)(r >> XB) << LC2) | coll2;
                const uint32_t r3 = ja2 | ((col >> LC3) << XA);
                const uint32_t t3 = (col & ((1u << LC3) - 1)) | (kc2 << LC3) | (kb << (LC3 + 3));
                lds_put<Fr>(xl, xh, r3 * PITCH + t3, v[r]);
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = lds_get<Fr>(xl, xh, r * PITCH + tid);
            // ---- stage 3: * w_D^(ja (kc + 8 kb)), DFT_RA over ja ------------------------------------
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t ja = r & (RA - 1);
                if (ja) v[r] = fp_mul_const29<Fr>(v[r], tw29_get<Fr>(T, launder((ja * (kc3 + (kb3 << 3))) << (8 - DB))));
            }
            ntt_dft_regs<Fr, XA>(v, T);
        }
        // ---- store ------------------------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if ((st_mask >> r) & 1u) {
                Fp<Fr> x = v[r];
                if ((tw_mask >> r) & 1u) x = fp_mul_const29<Fr>(x, tw[r]);
                fp_store<Fr>(dst + (size_t)st_off[r] * N, x);
            }
        }
        __syncthreads();  // LDS free for the next polynomial
    }
}

// ---- table construction ----------------------------------------------------------------------
template <class Fr>
struct RootConst;
template <>
struct RootConst<Bn254Fr> {
    static constexpr int TWO_ADICITY = 28;
    JF_HD static constexpr uint32_t ROOT(int i) {  // Fr::TWO_ADIC_ROOT_OF_UNITY, canonical integer
        constexpr uint32_t v[8] = {0x725b19f0u, 0x9bd61b6eu, 0x41112ed4u, 0x402d111eu,
                                   0x8ef62abcu, 0x00e0a7ebu, 0xa58a7e85u, 0x2a3c09f0u};
        return v[i];
    }
};
template <>
struct RootConst<Bls381Fr> {
    static constexpr int TWO_ADICITY = 32;
    JF_HD static constexpr uint32_t ROOT(int i) {
        constexpr uint32_t v[8] = {0x439f0d2bu, 0x3829971fu, 0x8c2280b9u, 0xb6368350u,
                                   0x22c813b4u, 0xd09b6819u, 0xdfe81f20u, 0x16a2a19eu};
        return v[i];
    }
};

// params: [0] w_n^(+-1), [1] w_256^(+-1), [2] n^-1, [3] h^(+-1)
template <class Fr>
__global__ void k_ntt_params(uint32_t log_n, int inverse, const uint32_t* __restrict__ coset, int has_coset,
                             uint32_t* __restrict__ params) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    constexpr int N = Fr::N;
    Fp<Fr> root;
    for (int i = 0; i < N; i++) root.v[i] = RootConst<Fr>::ROOT(i);
    root = fp_to_mont<Fr>(root);
    Fp<Fr> wn = root, w256 = root;
    for (uint32_t i = log_n; i < (uint32_t)RootConst<Fr>::TWO_ADICITY; i++) wn = fp_sqr<Fr>(wn);
    for (uint32_t i = 8; i < (uint32_t)RootConst<Fr>::TWO_ADICITY; i++) w256 = fp_sqr<Fr>(w256);
    Fp<Fr> h = has_coset ? fp_load<Fr>(coset) : Fp<Fr>::one();
    Fp<Fr> ninv = Fp<Fr>::one();
    if (inverse) {
        wn = fp_inv<Fr>(wn);
        w256 = fp_inv<Fr>(w256);
        h = fp_inv<Fr>(h);
        // n^-1 = (2^-1)^log_n
        Fp<Fr> two = fp_dbl<Fr>(Fp<Fr>::one());
        ninv = fp_pow_u64<Fr>(fp_inv<Fr>(two), log_n);
    }
    fp_store<Fr>(params, wn);
    fp_store<Fr>(params + N, w256);
    fp_store<Fr>(params + 2 * N, ninv);
    fp_store<Fr>(params + 3 * N, h);
}

// out[i] = scale * base^(i * stride)
template <class Fr>
__global__ __launch_bounds__(256) void k_pow_table(const uint32_t* __restrict__ base, const uint32_t* __restrict__ scale,
                                                   uint64_t stride, uint32_t count, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fp<Fr> v = fp_pow_u64<Fr>(fp_load<Fr>(base), (uint64_t)i * stride);
    if (scale) v = fp_mul<Fr>(v, fp_load<Fr>(scale));
    fp_store<Fr>(out + (size_t)i * Fr::N, v);
}

struct TableKey {
    int device, curve, inverse, has_coset;
    uint32_t log_n;
    uint64_t coset[4];
    bool operator<(const TableKey& o) const { return memcmp(this, &o, sizeof(TableKey)) < 0; }
};
static DevCache<TableKey, NttTables> g_tables;
void ntt_cache_clear() { g_tables.clear(); }

template <class Fr>
static int32_t build_tables(uint32_t log_n, int inverse, const uint64_t* coset_mont, NttTables* out,
                            jfhip_stream* st) {
    // allocations go straight into *out: on failure the cache releases whatever is already there
    constexpr int N = Fr::N;
    NttTables& t = *out;
    t.a = (log_n + 1) / 2;
    const uint32_t lo_n = 1u << t.a, hi_n = 1u << (log_n - t.a);
    const bool has_coset = coset_mont != nullptr;
    t.scale_mode = has_coset ? 2 : (inverse ? 1 : 0);
    t.nbytes = (size_t)((4 + 256 + (size_t)(has_coset ? 2 : 1) * (lo_n + hi_n)) * N * 4);
    JF_HIP(hipMalloc(&t.params, 4 * N * 4));
    JF_HIP(hipMalloc(&t.tw_small, 256 * N * 4));
    JF_HIP(hipMalloc(&t.tl, (size_t)lo_n * N * 4));
    JF_HIP(hipMalloc(&t.th, (size_t)hi_n * N * 4));
    uint32_t* d_coset = nullptr;
    if (has_coset) {
        JF_HIP(hipMalloc(&t.cl, (size_t)lo_n * N * 4));
        JF_HIP(hipMalloc(&t.ch, (size_t)hi_n * N * 4));
        d_coset = t.cl;  // the offset is staged in the first slot of its own power table, which the kernels below fill last
        JF_HIP(hipMemcpyAsync(d_coset, coset_mont, N * 4, hipMemcpyHostToDevice, st->s));
    }
    hipLaunchKernelGGL((k_ntt_params<Fr>), dim3(1), dim3(64), 0, st->s, log_n, inverse, d_coset, (int)has_coset,
                       t.params);
    hipLaunchKernelGGL((k_pow_table<Fr>), dim3(1), dim3(256), 0, st->s, t.params + N, (const uint32_t*)nullptr,
                       (uint64_t)1, 256u, t.tw_small);
    hipLaunchKernelGGL((k_pow_table<Fr>), dim3((lo_n + 255) / 256), dim3(256), 0, st->s, t.params,
                       (const uint32_t*)nullptr, (uint64_t)1, lo_n, t.tl);
    hipLaunchKernelGGL((k_pow_table<Fr>), dim3((hi_n + 255) / 256), dim3(256), 0, st->s, t.params,
                       (const uint32_t*)nullptr, (uint64_t)lo_n, hi_n, t.th);
    if (has_coset) {
        hipLaunchKernelGGL((k_pow_table<Fr>), dim3((hi_n + 255) / 256), dim3(256), 0, st->s, t.params + 3 * N,
                           inverse ? t.params + 2 * N : (const uint32_t*)nullptr, (uint64_t)lo_n, hi_n, t.ch);
        hipLaunchKernelGGL((k_pow_table<Fr>), dim3((lo_n + 255) / 256), dim3(256), 0, st->s, t.params + 3 * N,
                           (const uint32_t*)nullptr, (uint64_t)1, lo_n, t.cl);
    }
    JF_HIP(hipGetLastError());
    JF_HIP(hipStreamSynchronize(st->s));
    return JFHIP_OK;
}

// cached per (device, curve, log_n, direction, coset), bounded (DevCache)
template <class Fr>
static int32_t get_tables(int32_t curve, uint32_t log_n, int inverse, const uint64_t* coset_mont, NttTables* out,
                          jfhip_stream* st, Lease* lease) {
    TableKey key;
    memset(&key, 0, sizeof(key));
    JF_HIP(hipGetDevice(&key.device));
    key.curve = curve;
    key.inverse = inverse ? 1 : 0;
    key.has_coset = coset_mont ? 1 : 0;
    key.log_n = log_n;
    if (coset_mont) memcpy(key.coset, coset_mont, 32);
    const size_t hint = ((size_t)4 << ((log_n + 1) / 2)) * Fr::N * 4;
    return g_tables.get(key, out, lease, hint,
                        [&](NttTables* t) { return build_tables<Fr>(log_n, inverse ? 1 : 0, coset_mont, t, st); });
}

// two-level table of the forward domain generator: w^e = lo[e mod 2^a] * hi[e >> a]
int32_t ntt_root_tables(int32_t curve, uint32_t log_n, const uint32_t** lo, const uint32_t** hi, uint32_t* a,
                        jfhip_stream* st, Lease* lease) {
    NttTables t;
    if (curve == JFHIP_BN254) JF_TRY(get_tables<Bn254Fr>(curve, log_n, 0, nullptr, &t, st, lease));
    else JF_TRY(get_tables<Bls381Fr>(curve, log_n, 0, nullptr, &t, st, lease));
    *lo = t.tl;
    *hi = t.th;
    *a = t.a;
    return JFHIP_OK;
}

template <class Fr>
static int32_t ntt_device_t(int32_t curve, uint32_t* d_data, uint32_t log_n, size_t batch, int inverse,
                            const uint64_t* coset_mont, size_t in_len, jfhip_stream* st) {
    constexpr int N = Fr::N;
    if (log_n > (uint32_t)RootConst<Fr>::TWO_ADICITY || log_n > 30) {
        set_error("ntt: log_n %u exceeds the field's two-adicity / supported size", log_n);
        return JFHIP_ERR_INVALID;
    }
    const size_t n = (size_t)1 << log_n;
    if (in_len == 0 || in_len > n) in_len = n;
    if (batch == 0) return JFHIP_OK;
    if ((batch << log_n) >= ((size_t)1 << 40)) {
        set_error("ntt: batch too large");
        return JFHIP_ERR_INVALID;
    }
    // a trivial coset (h = 1) is the plain domain
    if (coset_mont) {
        bool is_one = true;
        for (int i = 0; i < 4; i++) {
            uint64_t one = ((uint64_t)Fr::ONE(2 * i + 1) << 32) | Fr::ONE(2 * i);
            if (coset_mont[i] != one) is_one = false;
        }
        if (is_one) coset_mont = nullptr;
    }
    if (log_n == 0) {
        // n = 1: the transform is the identity (h^0 = 1, n^-1 = 1); honour zero-extension only
        return JFHIP_OK;
    }
    PhaseTimer pt(st, 1);
    NttTables tabs;
    Lease lease;  // keeps the tables resident until every pass below is enqueued
    JF_TRY(get_tables<Fr>(curve, log_n, inverse ? 1 : 0, coset_mont, &tabs, st, &lease));
    // digit plan
    uint32_t ndig = (log_n + NTT_MAX_DIGIT - 1) / NTT_MAX_DIGIT;
    uint32_t dig[4] = {0, 0, 0, 0};
    for (uint32_t p = 0; p < ndig; p++) dig[p] = log_n / ndig + (p < log_n % ndig ? 1 : 0);

    // register-radix passes need >= 2 passes of 5..8 bits; tiny transforms keep the layer-per-exchange kernel
    static const bool force_v1 = getenv("JFHIP_NTT_V1") != nullptr;
    bool use_r8 = ndig >= 2 && !force_v1;
    for (uint32_t p = 0; p < ndig; p++) use_r8 = use_r8 && dig[p] >= 5 && dig[p] <= 8;

    uint32_t* tmp = nullptr;
    if (ndig >= 2) {
        JF_TRY(st->ntt_tmp.ensure(batch * n * N * 4));
        tmp = st->ntt_tmp.as<uint32_t>();
    }
    uint32_t log_M = log_n;
    for (uint32_t p = 0; p < ndig; p++) {
        NttPass P;
        P.first = (p == 0);
        P.final = (p == ndig - 1);
        P.inverse = inverse ? 1 : 0;
        P.src = (p == 0) ? d_data : tmp;
        P.dst = (ndig == 1 || P.final) ? d_data : tmp;
        P.log_n = log_n;
        // tiles must not straddle polynomials to loop over the batch inside a workgroup
        const bool loop_inside = (log_n - dig[p]) >= (NTT_TILE_LOG - dig[p]);
        P.outer = loop_inside ? 1u : (uint32_t)batch;
        P.inner = loop_inside ? (uint32_t)batch : 1u;
        P.dbits = dig[p];
        P.log_M = log_M;
        P.in_len = (uint32_t)in_len;
        P.ndig = ndig;
        for (int q = 0; q < 4; q++) P.dig[q] = dig[q];
        P.t = tabs;
        uint64_t ncols = (uint64_t)P.outer << (log_n - dig[p]);
        uint32_t logC = NTT_TILE_LOG - dig[p];
        uint64_t blocks = (ncols + (1ull << logC) - 1) >> logC;
        if (use_r8) {
            switch (dig[p]) {
#ifndef JF_NTT_QUICK
                case 5: hipLaunchKernelGGL((k_ntt_pass_r8<Fr, 5>), dim3((unsigned)blocks), dim3(NTT_THREADS), 0, st->s, P); break;
                case 6: hipLaunchKernelGGL((k_ntt_pass_r8<Fr, 6>), dim3((unsigned)blocks), dim3(NTT_THREADS), 0, st->s, P); break;
                case 7: hipLaunchKernelGGL((k_ntt_pass_r8<Fr, 7>), dim3((unsigned)blocks), dim3(NTT_THREADS), 0, st->s, P); break;
#endif
                default: hipLaunchKernelGGL((k_ntt_pass_r8<Fr, 8>), dim3((unsigned)blocks), dim3(NTT_THREADS), 0, st->s, P); break;
            }
        } else {
            hipLaunchKernelGGL((k_ntt_pass<Fr>), dim3((unsigned)blocks), dim3(NTT_THREADS), 0, st->s, P);
        }
        JF_HIP(hipGetLastError());
        pt.mark();
        log_M -= dig[p];
    }
    pt.finish();
    return JFHIP_OK;
}

int32_t ntt_device(int32_t curve, uint32_t* d_data, uint32_t log_n, size_t batch, int inverse,
                   const uint64_t* coset_mont, size_t in_len, jfhip_stream* st) {
    if (curve == JFHIP_BN254) return ntt_device_t<Bn254Fr>(curve, d_data, log_n, batch, inverse, coset_mont, in_len, st);
    return ntt_device_t<Bls381Fr>(curve, d_data, log_n, batch, inverse, coset_mont, in_len, st);
}

}  // namespace jf
