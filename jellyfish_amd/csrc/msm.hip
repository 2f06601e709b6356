// Pippenger multi-scalar multiplication for gfx950, the arithmetic behind
// UnivariateKzgPCS::commit / batch_commit / open (reference call sites
// pcs/src/univariate_kzg/mod.rs:114-118,166-170: E::G1::msm_bigint(..).into_affine()).
//
// MI355X-first design:
//   * the SRS handle holds table[w][i] = 2^(c*w) * P_i for every window w (288 GB of HBM
//     makes W copies of the SRS cheap), so all windows share ONE bucket space of
//     2^(c-1) buckets and no per-window Horner combine exists;
//   * scalars arrive in Montgomery form; the Montgomery reduction is fused into the
//     signed-digit decomposition (reference mod.rs:1044-1066 does it on the host);
//   * (bucket, table index) pairs are bucket-sorted on the device, then the sorted list is
//     cut into equal slices, one per lane, so every lane performs the same number of mixed
//     additions whatever the scalar distribution; runs that straddle a slice boundary are
//     stitched by a fix-up pass;
//   * buckets are folded by sum_k k*B_k = running sums per lane + a tree.
// All group arithmetic is exact (curve.hpp), so the affine result is the unique group
// element and equals arkworks' bit for bit.
#include <stdlib.h>

#include "common.hpp"
#include "curve.hpp"
#include "fp29.hpp"
#include "g29.hpp"

namespace jf {

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------

// bits [bit, bit+c) of an 8-limb integer, c <= 24, without dynamic register indexing
template <int N>
JF_DEV uint32_t extract_bits(const uint32_t (&v)[N], int bit, int c) {
    const int limb = bit >> 5, sh = bit & 31;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
        lo = (j == limb) ? v[j] : lo;
        hi = (j == limb + 1) ? v[j] : hi;
    }
    uint64_t t = ((uint64_t)hi << 32) | lo;
    return (uint32_t)(t >> sh) & ((1u << c) - 1u);
}

// ---- two-level bucket sort of (bucket, table index) pairs -------------------------------------
//
// bucket id = |digit| - 1 in [0, 2^(c-1)) is split into coarse (top) and fine (low) bits.
//   level 1: scalars -> signed digits (Montgomery reduction fused) -> entries appended to their
//            COARSE bin.  A workgroup handles a tile of scalars with an LDS histogram, reserves
//            its span of every coarse bin with ONE global atomic per bin, then scatters.
//   level 2: one workgroup per coarse bin counting-sorts its entries by the fine bits with LDS
//            atomics and emits the per-bucket offsets; its writes stay inside one L2-sized span.
// No per-entry global atomics, no digit array; entries cost 8 B between the levels.

constexpr int SORT_TILE_SCALARS = 2048;  // per workgroup of 256 lanes
constexpr int MAX_COARSE_BITS = 10;
constexpr int MAX_FINE_BITS = 18;  // consumed in one or two LDS levels of <= 9 bits

struct DigitPlan {
    int c, W, coarse_bits, fine_bits;
};

// signed digit of window w; carry in/out
template <int N>
JF_DEV int32_t signed_digit(const uint32_t (&v)[N], int w, int c, uint32_t& carry) {
    int bit = w * c;
    uint32_t raw = (bit < 32 * N) ? extract_bits<N>(v, bit, c) : 0u;
    raw += carry;
    const uint32_t half = 1u << (c - 1);
    if (raw > half) {
        carry = 1;
        return (int32_t)raw - (int32_t)(1u << c);
    }
    carry = 0;
    return (int32_t)raw;
}

// PASS = 0: coarse histogram only.  PASS = 1: reserve + scatter entries.  PASS = 2: per-tile histogram (for
// k_msm_coarse_staged).
// Which polynomial of a batch (MsmBatch, common.hpp) a sort tile belongs to: tiles are numbered polynomial by
// polynomial, mb.tile0[p] is the first tile of polynomial p.  Uniform over the workgroup.
JF_DEV uint32_t batch_entry_of_tile(const MsmBatch& mb, uint32_t tile) {
    uint32_t p = 0;
    for (uint32_t q = 1; q < mb.count; q++) p = (tile >= mb.tile0[q]) ? q : p;
    return p;
}

// A batch shares one sort: the bucket id of polynomial p is widened to p * 2^(c-1) + bucket.  The coarse level gets that
// for free -- a tile belongs to one polynomial, so its LDS histogram stays the 2^coarse_bits bins of ONE polynomial and
// only the global bin id carries p.
template <class Cv, int PASS>
__global__ __launch_bounds__(256) void k_msm_coarse(MsmBatch mb, DigitPlan pl, uint32_t tile_scalars, uint32_t table_n,
                                                    uint32_t* __restrict__ gcount /*PASS0: counts, PASS1: cursors*/,
                                                    uint2* __restrict__ entries) {
    using Fr = typename Cv::Fr;
    __shared__ uint32_t hist[1 << MAX_COARSE_BITS];
    __shared__ uint32_t base[1 << MAX_COARSE_BITS];
    const uint32_t nb = 1u << pl.coarse_bits;
    for (uint32_t k = threadIdx.x; k < nb; k += 256) hist[k] = 0;
    __syncthreads();
    const uint32_t pe = batch_entry_of_tile(mb, blockIdx.x);
    const uint32_t* __restrict__ scalars = mb.scalars[pe];
    const uint32_t n = mb.n[pe], offset = mb.offset[pe];
    const uint32_t ltile = blockIdx.x - mb.tile0[pe], ntiles = mb.tile0[pe + 1] - mb.tile0[pe];
    const uint32_t pbase = pe << (pl.c - 1);   // this polynomial's first global bucket id
    const uint32_t tile0 = ltile * tile_scalars;
    const int reps = (int)(tile_scalars / 256);

    for (int rep = 0; rep < reps; rep++) {
        uint32_t i = tile0 + rep * 256 + threadIdx.x;
        if (i < n) {
            Fp<Fr> s = fp_from_mont<Fr>(fp_load<Fr>(scalars + (size_t)i * Fr::N));
            uint32_t carry = 0;
            for (int w = 0; w < pl.W; w++) {
                int32_t d = signed_digit<Fr::N>(s.v, w, pl.c, carry);
                if (d != 0) {
                    uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
                    atomicAdd(&hist[b >> pl.fine_bits], 1u);
                }
            }
        }
    }
    __syncthreads();
    if (PASS == 0) {
        for (uint32_t k = threadIdx.x; k < nb; k += 256)
            if (hist[k]) atomicAdd(&gcount[pe * nb + k], hist[k]);
        return;
    }
    if (PASS == 2) {   // histogram of this tile, [polynomial][bin][tile] layout: its exclusive scan is every tile's offset in every bin
        for (uint32_t k = threadIdx.x; k < nb; k += 256)
            gcount[(size_t)nb * mb.tile0[pe] + (size_t)k * ntiles + ltile] = hist[k];
        return;
    }
    for (uint32_t k = threadIdx.x; k < nb; k += 256) {
        uint32_t h = hist[k];
        base[k] = h ? atomicAdd(&gcount[pe * nb + k], h) : 0u;
        hist[k] = 0;
    }
    __syncthreads();
    for (int rep = 0; rep < reps; rep++) {
        uint32_t i = tile0 + rep * 256 + threadIdx.x;
        if (i < n) {
            Fp<Fr> s = fp_from_mont<Fr>(fp_load<Fr>(scalars + (size_t)i * Fr::N));
            uint32_t carry = 0;
            for (int w = 0; w < pl.W; w++) {
                int32_t d = signed_digit<Fr::N>(s.v, w, pl.c, carry);
                if (d != 0) {
                    uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
                    uint32_t cb = b >> pl.fine_bits;
                    uint32_t pos = base[cb] + atomicAdd(&hist[cb], 1u);
                    uint32_t payload = ((uint32_t)w * table_n + offset + i) | (d < 0 ? 0x80000000u : 0u);
                    entries[pos] = make_uint2(payload, pbase | b);
                }
            }
        }
    }
}

// ---- refinement levels: equal-sized chunks of the entry array, any segment skew ----------------
//
// Input entries are grouped by the key bits above `shift + bits` (segments given by segbase).
// A workgroup takes a fixed chunk of LVL_CHUNK consecutive entries -- so the load is balanced
// whatever the bucket distribution (the top window alone puts 1/W of all entries into a handful
// of coarse bins) -- walks the segments the chunk overlaps in groups that fit the LDS histogram,
// and either counts (MODE 0) or reserves one span per (chunk, key) with a global atomic and
// scatters (MODE 1: entries, MODE 2: final payloads).  Keys of one level are at most 9 bits so a
// workgroup keeps few, L2-resident store frontiers.
constexpr int LVL_CHUNK = 8192;
constexpr int LVL_THREADS = 256;
constexpr int LVL_LDS_KEYS = 4096;

template <int MODE>
__global__ __launch_bounds__(LVL_THREADS) void k_msm_level(const uint2* __restrict__ in,
                                                           const uint32_t* __restrict__ segbase, uint32_t nseg,
                                                           int shift, int bits, uint32_t* __restrict__ gcount,
                                                           uint2* __restrict__ out_entries,
                                                           uint32_t* __restrict__ out_sorted) {
    __shared__ uint32_t cnt[LVL_LDS_KEYS];
    __shared__ uint32_t base[LVL_LDS_KEYS];
    const uint32_t total = segbase[nseg];
    const uint64_t c0_64 = (uint64_t)blockIdx.x * LVL_CHUNK;
    if (c0_64 >= total) return;
    const uint32_t c0 = (uint32_t)c0_64;
    const uint32_t c1 = (uint32_t)((c0_64 + LVL_CHUNK < total) ? c0_64 + LVL_CHUNK : total);
    const uint32_t seg_per_group = LVL_LDS_KEYS >> bits;
    // first segment overlapping the chunk: largest s with segbase[s] <= c0
    uint32_t lo = 0, hi = nseg;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (segbase[mid] <= c0) lo = mid;
        else hi = mid;
    }
    uint32_t s0 = lo;
    while (s0 < nseg) {
        uint32_t s1 = (s0 + seg_per_group < nseg) ? s0 + seg_per_group : nseg;
        uint32_t e_lo = segbase[s0], e_hi = segbase[s1];
        if (e_lo >= c1) break;
        if (e_lo < c0) e_lo = c0;
        if (e_hi > c1) e_hi = c1;
        if (e_hi > e_lo) {
            const uint32_t nkeys = (s1 - s0) << bits;
            for (uint32_t k = threadIdx.x; k < nkeys; k += LVL_THREADS) cnt[k] = 0;
            __syncthreads();
            // the bits above this level's key ARE the segment id, so the group-relative key is
            // one subtraction
            const uint32_t key0 = s0 << bits;
            for (uint32_t e = e_lo + threadIdx.x; e < e_hi; e += LVL_THREADS)
                atomicAdd(&cnt[(in[e].y >> shift) - key0], 1u);
            __syncthreads();
            uint32_t* g = gcount + ((size_t)s0 << bits);
            if (MODE == 0) {
                for (uint32_t k = threadIdx.x; k < nkeys; k += LVL_THREADS)
                    if (cnt[k]) atomicAdd(&g[k], cnt[k]);
            } else {
                for (uint32_t k = threadIdx.x; k < nkeys; k += LVL_THREADS) {
                    uint32_t h = cnt[k];
                    base[k] = h ? atomicAdd(&g[k], h) : 0u;
                    cnt[k] = 0;
                }
                __syncthreads();
                for (uint32_t e = e_lo + threadIdx.x; e < e_hi; e += LVL_THREADS) {
                    uint2 en = in[e];
                    uint32_t key = (en.y >> shift) - key0;
                    uint32_t pos = base[key] + atomicAdd(&cnt[key], 1u);
                    if (MODE == 2) out_sorted[pos] = en.x;
                    else out_entries[pos] = en;
                }
            }
            __syncthreads();
        }
        if (segbase[s1] >= c1) break;
        s0 = s1;
    }
}

// ---- staged scatters: a chunk is counting-sorted in LDS first, then copied out in runs ------------
//
// The direct scatters above store one 8-byte entry per lane to as many different places as the level has keys; with a
// span per (workgroup, key) the store frontiers of the resident workgroups do not fit the L2, so the lines go out
// partially written (k_msm_coarse<1> at 2^24: 2.1 GB moved in 2.14 ms, 68 % of the wave cycles waiting on memory).
// Here a workgroup ranks the entries of its chunk by key into 32 KiB of LDS and copies the staged array out in order:
// consecutive lanes store consecutive entries of one key's reserved span (runs of chunk / keys entries).  Every entry
// is read from global memory once and kept in registers between the counting and the ranking pass.
constexpr int STG_CAP = 4096;    // staged entries per flush: 32 KiB, four workgroups per CU
constexpr int STG_KEYS = 1024;   // LDS counters per flush (level kernels; the coarse level has <= 512 bins)
constexpr int STG_PER = STG_CAP / 256;

// cur[k] (counts on entry) -> start of key k inside the staged array; delta[k] = (span reserved in g[k]) - that start.
// Every lane of the 256-lane workgroup calls this; returns the number of staged entries.
// GLOBAL_G: g[] are global cursors shared by all workgroups (atomic reservation); otherwise g[] is owned by this
// workgroup (LDS) and key k is always handled by the same lane.
template <bool GLOBAL_G>
JF_DEV uint32_t stage_scan_reserve(uint32_t* cur, uint32_t* delta, uint32_t* wsum, uint32_t nkeys, uint32_t* g) {
    const uint32_t t = threadIdx.x, k0 = 4 * t, lane = t & 63u;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        v[i] = (k0 + i < nkeys) ? cur[k0 + i] : 0u;
        s += v[i];
    }
    uint32_t inc = s;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[t >> 6] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (t >> 6); w++) base += wsum[w];
    const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    uint32_t loc = base + inc - s;
    uint32_t gpos[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if constexpr (GLOBAL_G) gpos[i] = (k0 + i < nkeys && v[i]) ? atomicAdd(&g[k0 + i], v[i]) : 0u;
        else {
            gpos[i] = 0;
            if (k0 + i < nkeys) {
                gpos[i] = g[k0 + i];
                g[k0 + i] = gpos[i] + v[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (k0 + i < nkeys) {
            cur[k0 + i] = loc;
            delta[k0 + i] = gpos[i] - loc;
            loc += v[i];
        }
    }
    __syncthreads();
    return total;
}

// largest s in [0, nseg) with segbase[s] <= pos, by 64-way probes (three dependent loads for 2^18 segments instead of
// the eighteen of a binary search); all lanes of the wave return the same value.  Requires segbase[0] <= pos.
JF_DEV uint32_t seg_search64(const uint32_t* __restrict__ segbase, uint32_t nseg, uint32_t pos) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t lo = 0, len = nseg;   // answer in [lo, lo + len)
    while (len > 1) {
        const uint32_t step = (len + 63u) >> 6;
        const uint32_t idx = lo + lane * step;
        const bool le = idx < lo + len && segbase[idx] <= pos;
        const uint32_t cnt = (uint32_t)__popcll(__ballot(le));   // >= 1: lane 0 probes segbase[lo] <= pos
        const uint32_t nlo = lo + (cnt - 1u) * step;
        const uint32_t nlen = (nlo + step < lo + len) ? step : lo + len - nlo;
        lo = nlo;
        len = nlen;
    }
    return lo;
}

// k_msm_coarse<Cv, 1> with staging: 256 scalars (<= STG_CAP entries, W <= 16) per flush, digits kept in registers
template <class Cv>
__global__ __launch_bounds__(256) void k_msm_coarse_staged(MsmBatch mb, DigitPlan pl, uint32_t tile_scalars,
                                                           uint32_t table_n,
                                                           const uint32_t* __restrict__ tile_base /*[polynomial][bin][tile]*/,
                                                           uint2* __restrict__ entries) {
    using Fr = typename Cv::Fr;
    __shared__ uint2 stage[STG_CAP];
    __shared__ uint32_t cur[512], delta[512], tcur[512], wsum[4];
    const uint32_t nb = 1u << pl.coarse_bits;
    const uint32_t pe = batch_entry_of_tile(mb, blockIdx.x);
    const uint32_t* __restrict__ scalars = mb.scalars[pe];
    const uint32_t n = mb.n[pe], offset = mb.offset[pe];
    const uint32_t ltile = blockIdx.x - mb.tile0[pe], ntiles = mb.tile0[pe + 1] - mb.tile0[pe];
    const uint32_t pbase = pe << (pl.c - 1);
    // where this tile's entries of every bin go: exclusive scan of the [polynomial][bin][tile] histogram
    // (k_msm_coarse<Cv, 2>).  No atomics: with one global cursor per bin every flush of every workgroup queues on the same
    // <= 512 addresses (measured ~80 ns per same-address atomic: 65536 flushes = 4.9 ms at 2^24).
    for (uint32_t k = threadIdx.x; k < nb; k += 256)
        tcur[k] = tile_base[(size_t)nb * mb.tile0[pe] + (size_t)k * ntiles + ltile];
    const uint32_t tile0 = ltile * tile_scalars;
    const uint32_t tile1 = (tile0 + tile_scalars < n) ? tile0 + tile_scalars : n;
    for (uint32_t f0 = tile0; f0 < tile1; f0 += 256u) {
        const uint32_t i = f0 + threadIdx.x;
        // digit w as bucket | sign << 31, 0xffffffff = none
        uint32_t dg[16];
#pragma unroll
        for (int w = 0; w < 16; w++) dg[w] = 0xffffffffu;
        if (i < tile1) {
            Fp<Fr> s = fp_from_mont<Fr>(fp_load<Fr>(scalars + (size_t)i * Fr::N));
            uint32_t carry = 0;
#pragma unroll
            for (int w = 0; w < 16; w++) {
                if (w < pl.W) {
                    int32_t d = signed_digit<Fr::N>(s.v, w, pl.c, carry);
                    if (d != 0) dg[w] = ((uint32_t)(d < 0 ? -d : d) - 1u) | (d < 0 ? 0x80000000u : 0u);
                }
            }
        }
        for (uint32_t k = threadIdx.x; k < nb; k += 256) cur[k] = 0;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 16; w++)
            if (dg[w] != 0xffffffffu) atomicAdd(&cur[(dg[w] & 0x7fffffffu) >> pl.fine_bits], 1u);
        __syncthreads();
        const uint32_t m = stage_scan_reserve<false>(cur, delta, wsum, nb, tcur);
#pragma unroll
        for (int w = 0; w < 16; w++) {
            if (dg[w] != 0xffffffffu) {
                const uint32_t b = dg[w] & 0x7fffffffu;
                const uint32_t payload = ((uint32_t)w * table_n + offset + i) | (dg[w] & 0x80000000u);
                stage[atomicAdd(&cur[b >> pl.fine_bits], 1u)] = make_uint2(payload, b);
            }
        }
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < m; e += 256) {
            const uint2 en = stage[e];
            entries[delta[en.y >> pl.fine_bits] + e] = make_uint2(en.x, pbase | en.y);
        }
        __syncthreads();
    }
}

// k_msm_level<1 / 2> with staging (MODE 1: entries out, MODE 2: final payloads out); chunks of STG_CAP entries
template <int MODE>
__global__ __launch_bounds__(256) void k_msm_level_staged(const uint2* __restrict__ in,
                                                          const uint32_t* __restrict__ segbase, uint32_t nseg, int shift,
                                                          int bits, uint32_t* __restrict__ gcursor,
                                                          uint2* __restrict__ out_entries,
                                                          uint32_t* __restrict__ out_sorted) {
    __shared__ uint2 stage[STG_CAP];
    __shared__ uint32_t cur[STG_KEYS], delta[STG_KEYS], wsum[4];
    const uint32_t total = segbase[nseg];
    const uint64_t c0_64 = (uint64_t)blockIdx.x * STG_CAP;
    if (c0_64 >= total) return;
    const uint32_t c0 = (uint32_t)c0_64;
    const uint32_t c1 = (uint32_t)((c0_64 + STG_CAP < total) ? c0_64 + STG_CAP : total);
    const uint32_t seg_per_group = (uint32_t)STG_KEYS >> bits;
    // the chunk, read once: register k of lane t is entry c0 + k * 256 + t
    uint2 ent[STG_PER];
#pragma unroll
    for (int k = 0; k < STG_PER; k++) {
        const uint32_t e = c0 + k * 256u + threadIdx.x;
        ent[k] = e < c1 ? in[e] : make_uint2(0u, 0u);
    }
    uint32_t pos = c0;
    while (pos < c1) {
        // the segment that holds entry `pos`: empty stretches of segments are never walked
        const uint32_t s0 = seg_search64(segbase, nseg, pos);
        const uint32_t s1 = (s0 + seg_per_group < nseg) ? s0 + seg_per_group : nseg;
        uint32_t e_hi = segbase[s1];   // > pos: s1 > s0 and segment s0 holds pos
        if (e_hi > c1) e_hi = c1;
        const uint32_t nkeys = (s1 - s0) << bits;
        const uint32_t key0 = s0 << bits;
        for (uint32_t k = threadIdx.x; k < nkeys; k += 256) cur[k] = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < STG_PER; k++) {
            const uint32_t e = c0 + k * 256u + threadIdx.x;
            if (e >= pos && e < e_hi) atomicAdd(&cur[(ent[k].y >> shift) - key0], 1u);
        }
        __syncthreads();
        const uint32_t m = stage_scan_reserve<true>(cur, delta, wsum, nkeys, gcursor + ((size_t)s0 << bits));
#pragma unroll
        for (int k = 0; k < STG_PER; k++) {
            const uint32_t e = c0 + k * 256u + threadIdx.x;
            if (e >= pos && e < e_hi) stage[atomicAdd(&cur[(ent[k].y >> shift) - key0], 1u)] = ent[k];
        }
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < m; e += 256) {
            const uint2 en = stage[e];
            const uint32_t at = delta[(en.y >> shift) - key0] + e;
            if (MODE == 2) out_sorted[at] = en.x;
            else out_entries[at] = en;
        }
        __syncthreads();
        pos = e_hi;
    }
}

// ---- exclusive scan of up to 2^24 counters (3 small launches) ---------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_BLOCK = 256;

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_local(const uint32_t* __restrict__ in, uint32_t n,
                                                           uint32_t* __restrict__ out, uint32_t* __restrict__ bsum) {
    __shared__ uint32_t sh[SCAN_BLOCK];
    uint32_t base = (blockIdx.x * SCAN_BLOCK + threadIdx.x) * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t tot = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        tot += v[k];
    }
    sh[threadIdx.x] = tot;
    __syncthreads();
    for (int d = 1; d < SCAN_BLOCK; d <<= 1) {
        uint32_t t = (threadIdx.x >= (unsigned)d) ? sh[threadIdx.x - d] : 0u;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t excl = sh[threadIdx.x] - tot;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = excl;
        excl += v[k];
    }
    if (threadIdx.x == SCAN_BLOCK - 1) bsum[blockIdx.x] = sh[threadIdx.x];
}

// single block: exclusive scan of the block sums in place; total -> out_total
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_top(uint32_t* __restrict__ bsum, uint32_t nblk,
                                                         uint32_t* __restrict__ out_total) {
    __shared__ uint32_t sh[SCAN_BLOCK];
    uint32_t per = (nblk + SCAN_BLOCK - 1) / SCAN_BLOCK;
    uint32_t lo = threadIdx.x * per;
    uint32_t tot = 0;
    for (uint32_t k = 0; k < per; k++)
        if (lo + k < nblk) tot += bsum[lo + k];
    sh[threadIdx.x] = tot;
    __syncthreads();
    for (int d = 1; d < SCAN_BLOCK; d <<= 1) {
        uint32_t t = (threadIdx.x >= (unsigned)d) ? sh[threadIdx.x - d] : 0u;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t excl = sh[threadIdx.x] - tot;
    for (uint32_t k = 0; k < per; k++)
        if (lo + k < nblk) {
            uint32_t t = bsum[lo + k];
            bsum[lo + k] = excl;
            excl += t;
        }
    if (threadIdx.x == SCAN_BLOCK - 1) *out_total = sh[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_scan_add(uint32_t* __restrict__ off, uint32_t n,
                                                  const uint32_t* __restrict__ bsum, uint32_t* __restrict__ cursor) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v = off[i] + bsum[i / (SCAN_BLOCK * SCAN_ITEMS)];
    off[i] = v;
    if (cursor) cursor[i] = v;
}

// coarse segment bases from the scanned [polynomial][bin][tile] histogram: cbase[p * nb + k] = scanned[first tile of
// (p, k)], cbase[count * nb] = total = off[B]  (scanned has one more entry than counters: the total)
__global__ __launch_bounds__(256) void k_msm_coarse_bases(const uint32_t* __restrict__ scanned, uint32_t nb, MsmBatch mb,
                                                          uint32_t* __restrict__ cbase, uint32_t* __restrict__ off_total) {
    const uint32_t kg = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nbg = nb * mb.count;
    if (kg > nbg) return;
    const uint32_t pe = kg == nbg ? mb.count : kg / nb, k = kg == nbg ? 0u : kg % nb;
    // an empty polynomial (no tiles) has no counters of its own: its bins start where the next polynomial's do
    const uint32_t ntiles = kg == nbg ? 0u : mb.tile0[pe + 1] - mb.tile0[pe];
    const uint32_t v = scanned[(size_t)nb * mb.tile0[pe] + (size_t)k * ntiles];
    cbase[kg] = v;
    if (kg == nbg) *off_total = v;
}

// Advance (b, b_lo, b_hi) to the bucket that holds sorted position `pos` (pos >= b_hi on entry).  The next bucket is
// the usual answer; when it is empty the rest is a binary search, not a walk: a sparse polynomial (an unused wire is
// four blinding coefficients among 2^20 zeros) leaves tens of thousands of empty buckets between two entries, and
// walking them with one dependent load each cost 5.7 ms per such commitment (2 of the 5 wire polynomials of the bench
// circuit).
JF_DEV void next_bucket(const uint32_t* __restrict__ off, uint32_t B, uint32_t pos, uint32_t& b, uint32_t& b_lo,
                        uint32_t& b_hi) {
    b++;
    b_lo = b_hi;
    b_hi = off[b + 1];
    if (pos < b_hi) return;
    // largest bucket index with off[.] <= pos; off is non-decreasing and off[B] = total > pos
    uint32_t lo = b + 1, hi = B;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= pos) lo = mid;
        else hi = mid;
    }
    b = lo;
    b_lo = off[b];
    b_hi = off[b + 1];
}

// One lane per slice of L consecutive sorted entries.  A run (maximal stretch of one bucket
// inside a slice) that covers its whole bucket is final and goes to buckets[b]; a run cut
// by a slice boundary goes to partA[t] (it contains the slice's first entry) or partB[t]
// (it does not) and is stitched by k_msm_fixup.
// Slice length of a call.  The host sizes T slices (one per lane) for the most entries the call can have, n x windows; how
// many entries exist -- the non-zero digits -- is known on the device only (off[B] after the sort).  Every kernel after the
// sort derives the SAME length from off[B] and T, so that with sparse scalars (witness values of a few bits fill one or two
// of the 12-16 windows) all T lanes still get an equal share instead of the first few getting full-length slices.
// lmin (16, or JFHIP_MSM_MIN_SLICE) bounds the per-slice overhead (a bucket search and two partial sums per lane).
struct SliceCfg {
    uint32_t T;      // slices the workspace holds
    uint32_t lmin;   // shortest slice
};
JF_DEV uint32_t msm_slice_len(uint32_t total, SliceCfg sc) {
    const uint32_t l = (uint32_t)(((uint64_t)total + sc.T - 1) / sc.T);
    return l < sc.lmin ? sc.lmin : l;
}

template <class Cv>
__global__ __launch_bounds__(128) void k_msm_accumulate(const uint32_t* __restrict__ sorted,
                                                        const uint32_t* __restrict__ off, uint32_t B,
                                                        const uint32_t* __restrict__ table, SliceCfg T,
                                                        uint32_t* __restrict__ buckets, uint32_t* __restrict__ partA,
                                                        uint32_t* __restrict__ partB) {
    using Q = typename Cv::Fq;
    constexpr int PT = 2 * Q::N;  // limbs per packed affine point
    constexpr int XZ = 4 * Q::N;  // limbs per XYZZ point
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = off[B];
    const uint32_t L = msm_slice_len(total, T);
    const uint64_t start64 = (uint64_t)t * L;
    if (start64 >= total) return;
    const uint32_t start = (uint32_t)start64;
    const uint32_t end = (uint32_t)((start64 + L < total) ? start64 + L : total);

    // bucket containing `start`: largest b with off[b] <= start (and off[b+1] > start)
    uint32_t lo = 0, hi = B;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= start) lo = mid;
        else hi = mid;
    }
    uint32_t b = lo;
    uint32_t b_lo = off[b], b_hi = off[b + 1];
    while (b_hi <= start) {  // skip empty buckets sharing the same offset
        b++;
        b_lo = b_hi;
        b_hi = off[b + 1];
    }

    XYZZ<Cv> acc = XYZZ<Cv>::identity();
    uint32_t run_start = start;
    bool wroteA = false, wroteB = false;

    auto flush = [&](uint32_t run_end) {
        bool complete = (run_start == b_lo) && (run_end == b_hi);
        uint32_t* dst;
        if (complete) dst = buckets + (size_t)b * XZ;
        else if (run_start == start) {
            dst = partA + (size_t)t * XZ;
            wroteA = true;
        } else {
            dst = partB + (size_t)t * XZ;
            wroteB = true;
        }
        xyzz_store<Cv>(dst, acc);
    };

    for (uint32_t pos = start; pos < end; pos++) {
        if (pos >= b_hi) {
            flush(pos);
            acc = XYZZ<Cv>::identity();
            run_start = pos;
            next_bucket(off, B, pos, b, b_lo, b_hi);
        }
        uint32_t e = sorted[pos];
        Affine<Cv> p = affine_load<Cv>(table + (size_t)(e & 0x7fffffffu) * PT);
        if (e >> 31) p.y = fp_neg<Q>(p.y);
        xyzz_add_affine<Cv>(acc, p);
    }
    flush(end);
    if (!wroteA) xyzz_store<Cv>(partA + (size_t)t * XZ, XYZZ<Cv>::identity());
    if (!wroteB) xyzz_store<Cv>(partB + (size_t)t * XZ, XYZZ<Cv>::identity());
}

// ---- the same accumulation on unsaturated 29-bit limbs (fp29.hpp) ---------------------------------------
// The window table is stored with the Montgomery factor R' = 2^(29 L) (srs->f29), so a table point only needs its bits
// re-sliced.  Lazy-reduction bounds, in multiples of p (BN254 Fq, the tightest field: 2^257 = 10.58 p is the contraction
// limit of f29_mul, whose outputs are < 1.66 p):  X < 7.66, Y < 3.66, ZZ, ZZZ < 1.66;  P = U2 - X + 8p < 9.66,
// R = S2 - Y + 4p < 5.66, X3 = R^2 - PPP - 2Q + 6p < 7.66, Q - X3 + 8p < 9.66, Y3 = .. - Y PPP + 2p < 3.66.
template <class Cv>
JF_DEV void xyzz29_store_raw(uint32_t* __restrict__ dst, const XYZZ29<Cv>& a, bool inf) {
    constexpr int L = F29<typename Cv::Fq>::L;
    static_assert((4 * L) % 4 == 0, "");
    uint32_t w[4 * L];
#pragma unroll
    for (int k = 0; k < L; k++) {
        w[k] = inf ? 0u : a.X.v[k];
        w[L + k] = inf ? 0u : a.Y.v[k];
        w[2 * L + k] = inf ? 0u : a.ZZ.v[k];
        w[3 * L + k] = inf ? 0u : a.ZZZ.v[k];
    }
#pragma unroll
    for (int k = 0; k < L; k++)
        reinterpret_cast<uint4*>(dst)[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}

template <class Cv>
JF_DEV bool xyzz29_load_raw(const uint32_t* __restrict__ src, XYZZ29<Cv>& a) {   // returns "is the identity"
    constexpr int L = F29<typename Cv::Fq>::L;
    uint32_t w[4 * L];
#pragma unroll
    for (int k = 0; k < L; k++) {
        const uint4 q = reinterpret_cast<const uint4*>(src)[k];
        w[4 * k] = q.x;
        w[4 * k + 1] = q.y;
        w[4 * k + 2] = q.z;
        w[4 * k + 3] = q.w;
    }
    uint32_t nz = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
        a.X.v[k] = w[k];
        a.Y.v[k] = w[L + k];
        a.ZZ.v[k] = w[2 * L + k];
        a.ZZZ.v[k] = w[3 * L + k];
        nz |= w[2 * L + k];
    }
    return nz == 0;
}
// rare path of the mixed add (P + P or P - P): fall back to the saturated formulas, out of line.  Everything goes
// in and out BY VALUE: a reference would make the hot loop's accumulator escape into scratch memory.
template <class Cv>
struct Rare29 {
    XYZZ29<Cv> acc;
    bool inf;
};
// The point comes in as its table address and sign, not as the limbs the hot loop holds: those then die right after
// their product instead of staying live (18 registers) until the P == 0 test.
template <class Cv>
JF_COLD Rare29<Cv> xyzz29_add_rare(XYZZ29<Cv> acc, const uint32_t* __restrict__ table_point, bool negate) {
    using Q = typename Cv::Fq;
    XYZZ<Cv> a = xyzz29_to_std<Cv>(acc, false);
    Affine<Cv> q = affine_load<Cv>(table_point);   // bits of (x R', y R')
    q.x = f29_to_fp<Q>(f29_unpack_bits<Q>(q.x));
    q.y = f29_to_fp<Q>(f29_unpack_bits<Q>(q.y));
    if (negate) q.y = fp_neg<Q>(q.y);
    xyzz_add_affine<Cv>(a, q);
    Rare29<Cv> r;
    r.inf = a.is_inf();
    r.acc.X = f29_from_fp<Q>(a.X);
    r.acc.Y = f29_from_fp<Q>(a.Y);
    r.acc.ZZ = f29_from_fp<Q>(a.ZZ);
    r.acc.ZZZ = f29_from_fp<Q>(a.ZZZ);
    return r;
}

template <class Cv, int OCC>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void k_msm_accumulate29(const uint32_t* __restrict__ sorted,
                                                          const uint32_t* __restrict__ off, uint32_t B,
                                                          const uint32_t* __restrict__ table, SliceCfg T,
                                                          uint32_t* __restrict__ buckets, uint32_t* __restrict__ partA,
                                                          uint32_t* __restrict__ partB) {
    // buckets / partA / partB are RAW here: 4 L29 words per element, lazily reduced 29-bit limbs, all zero = identity.
    // Converting to the saturated form inside flush() would put ~1400 instructions on a branch that some lane of
    // a wave takes in half of all iterations (average bucket ~ 96 entries, 64 lanes); the fix-up, the fold and the
    // sum tree add them in this form (Rep<Cv, true>); only the MSM's one result is converted.
    using Q = typename Cv::Fq;
    using F = F29<Q>;
    constexpr int PT = 2 * Q::N;
    constexpr int XZ = 4 * F::L;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = off[B];
    const uint32_t L = msm_slice_len(total, T);
    const uint64_t start64 = (uint64_t)t * L;
    if (start64 >= total) return;
    const uint32_t start = (uint32_t)start64;
    const uint32_t end = (uint32_t)((start64 + L < total) ? start64 + L : total);

    uint32_t lo = 0, hi = B;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= start) lo = mid;
        else hi = mid;
    }
    uint32_t b = lo;
    uint32_t b_lo = off[b], b_hi = off[b + 1];
    while (b_hi <= start) {
        b++;
        b_lo = b_hi;
        b_hi = off[b + 1];
    }

    // end of the FOLLOWING bucket, fetched one run ahead: the hand-over to the next bucket then needs no load it has to
    // wait for (off has B + 1 entries; b + 1 == B only for the last bucket, whose successor is never entered)
    uint32_t nx_hi = off[b + 2 <= B ? b + 2 : B];
    XYZZ29<Cv> acc;
    bool inf = true;
    uint32_t run_start = start;
    bool wroteA = false, wroteB = false;
    F one29 = f29_from_fp<Q>(Fp<Q>::one());
    F zero29;
#pragma unroll
    for (int i = 0; i < F::L; i++) zero29.v[i] = 0;

    auto flush = [&](uint32_t run_end) {
        bool complete = (run_start == b_lo) && (run_end == b_hi);
        uint32_t* dst;
        if (complete) dst = buckets + (size_t)b * XZ;
        else if (run_start == start) {
            dst = partA + (size_t)t * XZ;
            wroteA = true;
        } else {
            dst = partB + (size_t)t * XZ;
            wroteB = true;
        }
        xyzz29_store_raw<Cv>(dst, acc, inf);
    };

    // software pipeline: the table point of entry pos + 1 (a random 64 / 96-byte gather from a multi-GiB table) and the
    // sorted entry pos + 2 are in flight while entry pos is added.  Order matters: vmcnt retires in order, so the bucket
    // hand-over (dependent loads on off[], waited for at once) comes BEFORE the gather is issued -- after it, every such
    // wait also waited for the gather (half of all iterations have some lane at a bucket boundary).  The gathered words
    // pass through an empty asm at the bottom of the iteration, their only use: with the uses visible the compiler
    // re-fetches words it wants as register pairs with overlapping loads (7 loads, 22 registers for 64 bytes).
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int NV = PT / 4;
    auto gather = [&](uint32_t e, u32x4* dst) {
        const u32x4* src = reinterpret_cast<const u32x4*>(table + (size_t)(e & 0x7fffffffu) * PT);
#pragma unroll
        for (int k = 0; k < NV; k++) dst[k] = src[k];
    };
    uint32_t e_cur = sorted[start];
    uint32_t e_nxt = (start + 1 < end) ? sorted[start + 1] : e_cur;
    u32x4 cur[NV], nxt[NV];
    gather(e_cur, cur);
    for (uint32_t pos = start; pos < end; pos++) {
        if (pos >= b_hi) {
            flush(pos);
            inf = true;
            run_start = pos;
            b++;
            b_lo = b_hi;
            b_hi = nx_hi;
            if (pos >= b_hi) {   // empty buckets follow: binary search for the one holding pos (see next_bucket)
                uint32_t slo = b + 1, shi = B;
                while (shi - slo > 1) {
                    uint32_t mid = (slo + shi) >> 1;
                    if (off[mid] <= pos) slo = mid;
                    else shi = mid;
                }
                b = slo;
                b_lo = off[b];
                b_hi = off[b + 1];
            }
            nx_hi = off[b + 2 <= B ? b + 2 : B];
        }
        gather(e_nxt, nxt);
        const uint32_t e_nx2 = (pos + 2 < end) ? sorted[pos + 2] : e_nxt;
        do {
            const uint32_t e = e_cur;
            Affine<Cv> p;
#pragma unroll
            for (int k = 0; k < NV; k++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int w = 4 * k + j;
                    if (w < Q::N) p.x.v[w] = cur[k][j];
                    else p.y.v[w - Q::N] = cur[k][j];
                }
            }
            if (p.is_inf()) break;
            F x2 = f29_unpack_bits<Q>(p.x);
            F y2 = f29_unpack_bits<Q>(p.y);
            if (e >> 31) y2 = f29_sub<Q, 1>(zero29, y2);   // p - y (y < p)
            if (inf) {
                acc.X = x2;
                acc.Y = y2;
                acc.ZZ = one29;
                acc.ZZZ = one29;
                inf = false;
                break;
            }
            F U2 = f29_mul<Q>(x2, acc.ZZ);
            F S2 = f29_mul<Q>(y2, acc.ZZZ);
            F P = f29_sub<Q, 8>(U2, acc.X);
            F PP = f29_sqr<Q>(P);
            if (f29_mulout_is_zero<Q>(PP)) {
                XYZZ29<Cv> ac = acc;
                Rare29<Cv> rr = xyzz29_add_rare<Cv>(ac, table + (size_t)(e & 0x7fffffffu) * PT, (e >> 31) != 0);
                acc = rr.acc;
                inf = rr.inf;
                break;
            }
            F R = f29_sub<Q, 4>(S2, acc.Y);
            F PPP = f29_mul<Q>(P, PP);
            F Qv = f29_mul<Q>(acc.X, PP);
            F X3 = f29_sub3<Q, 6>(f29_sqr<Q>(R), PPP, Qv, Qv);
            // Y3 = R (Q - X3) - Y PPP as ONE reduction: R (Q - X3 + 8p) + (4p - Y) PPP < (5.66 * 9.66 + 4 * 1.66) p^2 / R' + p
            // < 1.37 p (p / R' < 1 / 168 for every supported field)
            F Y3 = f29_dot2<Q>(R, f29_sub<Q, 8>(Qv, X3), f29_sub<Q, 4>(zero29, acc.Y), PPP);
            acc.X = X3;
            acc.Y = Y3;
            acc.ZZ = f29_mul<Q>(acc.ZZ, PP);
            acc.ZZZ = f29_mul<Q>(acc.ZZZ, PPP);
        } while (0);
#pragma unroll
        for (int k = 0; k < NV; k++) {
            asm volatile("" : "+v"(nxt[k]));
            cur[k] = nxt[k];
        }
        e_cur = e_nxt;
        e_nxt = e_nx2;
    }
    flush(end);
    if (!wroteA) xyzz29_store_raw<Cv>(partA + (size_t)t * XZ, acc, true);
    if (!wroteB) xyzz29_store_raw<Cv>(partB + (size_t)t * XZ, acc, true);
}

// table entries x -> x 2^(+-SHIFT): saturated Montgomery factor R <-> R' (identity entries (0, 0) are unchanged)
template <class Cv>
__global__ __launch_bounds__(256) void k_table_rescale(uint32_t* __restrict__ pts, size_t count, int up) {
    using Q = typename Cv::Fq;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * count) return;
    Fp<Q> x = fp_load<Q>(pts + i * Q::N);
    x = up ? fp_scale_up_r29<Q>(x) : fp_scale_down_r29<Q>(x);
    fp_store<Q>(pts + i * Q::N, x);
}

int32_t table_rescale(int32_t curve, uint32_t* d_pts, size_t count, int up, hipStream_t s) {
    if (count == 0) return JFHIP_OK;
    dim3 grid((unsigned)((2 * count + 255) / 256));
    if (curve == JFHIP_BN254) hipLaunchKernelGGL((k_table_rescale<Bn254G1>), grid, dim3(256), 0, s, d_pts, count, up);
    else hipLaunchKernelGGL((k_table_rescale<Bls381G1>), grid, dim3(256), 0, s, d_pts, count, up);
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

// ---- group arithmetic of the stages after the accumulation, in either representation -------------------------
// The accumulation's runs are RAW (29-bit limbs, lazily reduced, the invariants of k_msm_accumulate29: X < 7.66 p,
// Y < 3.66 p, ZZ, ZZZ < 1.66 p; all zero = identity).  The fix-up, the fold and the sum tree add them as they are: a full
// XYZZ addition on the 29-bit multiplier is ~3 300 instructions against ~5 200 saturated, and no element is converted
// before the one result of the MSM.
// the two representations behind one interface: RAW = false saturated XYZZ (4 N words), RAW = true the above (4 L words)
template <class Cv, bool RAW>
struct Rep;
template <class Cv>
struct Rep<Cv, false> {
    using T = XYZZ<Cv>;
    static constexpr int W = 4 * Cv::Fq::N;
    JF_DEV static T load(const uint32_t* __restrict__ base, size_t i) { return xyzz_load<Cv>(base + i * W); }
    JF_DEV static void store(uint32_t* __restrict__ base, size_t i, const T& a) { xyzz_store<Cv>(base + i * W, a); }
    JF_DEV static T identity() { return XYZZ<Cv>::identity(); }
    JF_DEV static void add(T& acc, const T& q) { xyzz_add_inl<Cv>(acc, q); }
    JF_DEV static T dbl(const T& a) { return xyzz_double<Cv>(a); }
    JF_DEV static XYZZ<Cv> to_std(const T& a) { return a; }
};
template <class Cv>
struct Rep<Cv, true> {
    using T = P29<Cv>;
    static constexpr int W = 4 * F29<typename Cv::Fq>::L;
    JF_DEV static T load(const uint32_t* __restrict__ base, size_t i) {
        T a;
        a.inf = xyzz29_load_raw<Cv>(base + i * W, a.v);
        return a;
    }
    JF_DEV static void store(uint32_t* __restrict__ base, size_t i, const T& a) {
        xyzz29_store_raw<Cv>(base + i * W, a.v, a.inf);
    }
    JF_DEV static T identity() {
        T a;
        a.inf = true;
#pragma unroll
        for (int k = 0; k < F29<typename Cv::Fq>::L; k++) a.v.X.v[k] = a.v.Y.v[k] = a.v.ZZ.v[k] = a.v.ZZZ.v[k] = 0;
        return a;
    }
    JF_DEV static void add(T& acc, const T& q) { p29_add<Cv>(acc, q); }
    JF_DEV static T dbl(const T& a) { return p29_double<Cv>(a); }
    JF_DEV static XYZZ<Cv> to_std(const T& a) { return xyzz29_to_std<Cv>(a.v, a.inf); }
};

// buckets spanning >= 2 slices: sum their pieces.  One lane per bucket; a bucket spanning more than HEAVY_SPAN slices
// (skewed scalars: all-equal scalars put n entries into one bucket per window = 65536 slices at 2^24, 0.8 s of serial
// additions on one lane) is queued for k_msm_fixup_heavy instead.
constexpr uint32_t HEAVY_SPAN = 48;
// partA / partB: what the accumulation wrote; buckets: written here for the spanning buckets only, in the same
// representation -- k_msm_bucket_fold takes every other bucket from the accumulation's own array.
template <class Cv, bool RAW>
__global__ __launch_bounds__(128) void k_msm_fixup(const uint32_t* __restrict__ off, uint32_t B, SliceCfg T,
                                                   const uint32_t* __restrict__ partA,
                                                   const uint32_t* __restrict__ partB,
                                                   uint32_t* __restrict__ buckets, uint32_t* __restrict__ heavy) {
    using G = Rep<Cv, RAW>;
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    uint32_t s = off[b], e = off[b + 1];
    if (e <= s) return;
    const uint32_t L = msm_slice_len(off[B], T);
    uint32_t t1 = s / L, t2 = (e - 1) / L;
    if (t1 == t2) return;
    if (t2 - t1 > HEAVY_SPAN) {
        heavy[1 + atomicAdd(heavy, 1u)] = b;
        return;
    }
    typename G::T acc = G::load((s % L == 0) ? partA : partB, t1);
    for (uint32_t t = t1 + 1; t <= t2; t++) {
        typename G::T q = G::load(partA, t);
        G::add(acc, q);
    }
    G::store(buckets, b, acc);
}

// Queued (heavy) buckets, two launches.  A bucket with all-equal or small-valued scalars holds up to n entries: tens of
// thousands of slices once the slices are short (msm_slice_len), too many for one workgroup's 256 lanes to add up in
// sequence (3.6 ms for a 2^22-entry bucket of BLS12-381).  k_msm_fixup_heavy_chunks cuts a bucket's full slices
// t1 + 1 .. t2 into chunks of HEAVY_CHUNK, one workgroup per chunk (dealt round-robin over the grid), and leaves the chunk's sum IN PLACE in partA at the chunk's first slice -- the slices of a
// chunk belong to this bucket alone and are read only by the workgroup that owns the chunk, before it writes.
// k_msm_fixup_heavy then adds the bucket's first piece and its chunk sums, one workgroup per bucket.
constexpr uint32_t HEAVY_CHUNK = 1024;
template <class G>
JF_DEV typename G::T heavy_block_sum(uint32_t* sh, typename G::T acc) {   // sum over the 256 lanes, valid in lane 0
    G::store(sh, threadIdx.x, acc);
    __syncthreads();
    for (uint32_t w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) {
            typename G::T q = G::load(sh, threadIdx.x + w);
            G::add(acc, q);
            G::store(sh, threadIdx.x, acc);
        }
        __syncthreads();
    }
    return acc;
}
template <class Cv, bool RAW>
__global__ __launch_bounds__(256) void k_msm_fixup_heavy_chunks(const uint32_t* __restrict__ off, uint32_t B, SliceCfg T,
                                                                uint32_t* __restrict__ partA,
                                                                const uint32_t* __restrict__ heavy) {
    using G = Rep<Cv, RAW>;
    __shared__ uint32_t sh[256 * G::W];
    __shared__ uint32_t q_t1[256], q_t2[256];
    const uint32_t count = heavy[0];
    const uint32_t L = msm_slice_len(off[B], T);
    uint32_t qbase = 0;
    for (uint32_t base = 0; base < count; base += 256u) {
        __syncthreads();
        if (base + threadIdx.x < count) {
            const uint32_t b = heavy[1 + base + threadIdx.x];
            q_t1[threadIdx.x] = off[b] / L;
            q_t2[threadIdx.x] = (off[b + 1] - 1) / L;
        }
        __syncthreads();
        const uint32_t m = count - base < 256u ? count - base : 256u;
        for (uint32_t j = 0; j < m; j++) {
            const uint32_t t1 = q_t1[j], t2 = q_t2[j];
            const uint32_t chunks = (t2 - t1 + HEAVY_CHUNK - 1) / HEAVY_CHUNK;   // slices t1 + 1 .. t2
            if (chunks <= 1) continue;   // a single chunk is summed by k_msm_fixup_heavy directly
            // chunks are numbered through the whole queue (every workgroup keeps the same running count) and dealt
            // round-robin: neighbouring buckets must not land on the same few workgroups
            uint32_t c = (blockIdx.x + gridDim.x - qbase % gridDim.x) % gridDim.x;
            qbase += chunks;
            for (; c < chunks; c += gridDim.x) {
                const uint32_t first = t1 + 1 + c * HEAVY_CHUNK;
                const uint32_t last = first + HEAVY_CHUNK - 1 < t2 ? first + HEAVY_CHUNK - 1 : t2;
                typename G::T acc = G::identity();
                for (uint32_t t = first + threadIdx.x; t <= last; t += 256u) {
                    typename G::T q = G::load(partA, t);
                    G::add(acc, q);
                }
                acc = heavy_block_sum<G>(sh, acc);   // every lane's loads are done at its first barrier
                if (threadIdx.x == 0) G::store(partA, first, acc);
            }
        }
    }
}

template <class Cv, bool RAW>
__global__ __launch_bounds__(256) void k_msm_fixup_heavy(const uint32_t* __restrict__ off, uint32_t B, SliceCfg T,
                                                         const uint32_t* __restrict__ partA,
                                                         const uint32_t* __restrict__ partB,
                                                         uint32_t* __restrict__ buckets,
                                                         const uint32_t* __restrict__ heavy) {
    using G = Rep<Cv, RAW>;
    __shared__ uint32_t sh[256 * G::W];
    const uint32_t count = heavy[0];
    const uint32_t L = msm_slice_len(off[B], T);
    for (uint32_t h = blockIdx.x; h < count; h += gridDim.x) {
        const uint32_t b = heavy[1 + h];
        const uint32_t s = off[b], e = off[b + 1];
        const uint32_t t1 = s / L, t2 = (e - 1) / L;
        const uint32_t chunks = (t2 - t1 + HEAVY_CHUNK - 1) / HEAVY_CHUNK;
        typename G::T acc = G::identity();
        if (threadIdx.x == 0) acc = G::load((s % L == 0) ? partA : partB, t1);
        if (chunks <= 1) {
            for (uint32_t t = t1 + 1 + threadIdx.x; t <= t2; t += 256u) {
                typename G::T q = G::load(partA, t);
                G::add(acc, q);
            }
        } else {
            for (uint32_t c = threadIdx.x; c < chunks; c += 256u) {   // chunk sums left in place by k_msm_fixup_heavy_chunks
                typename G::T q = G::load(partA, t1 + 1 + c * HEAVY_CHUNK);
                G::add(acc, q);
            }
        }
        acc = heavy_block_sum<G>(sh, acc);
        if (threadIdx.x == 0) G::store(buckets, b, acc);
        __syncthreads();
    }
}

// k * p for a small non-negative integer k (double-and-add, MSB first)
template <class Cv>
JF_DEV XYZZ<Cv> xyzz_mul_small(const XYZZ<Cv>& p, uint32_t k) {
    XYZZ<Cv> r = XYZZ<Cv>::identity();
    for (int bit = 31 - __clz(k | 1u); bit >= 0; bit--) {
        r = xyzz_double<Cv>(r);
        if ((k >> bit) & 1u) xyzz_add<Cv>(r, p);
    }
    return r;
}
template <class Cv, bool RAW>
JF_DEV typename Rep<Cv, RAW>::T rep_mul_small(const typename Rep<Cv, RAW>::T& p, uint32_t k) {
    using G = Rep<Cv, RAW>;
    typename G::T r = G::identity();
    for (int bit = 31 - __clz(k | 1u); bit >= 0; bit--) {
        r = G::dbl(r);
        if ((k >> bit) & 1u) G::add(r, p);
    }
    return r;
}

// lane t folds buckets [t*g, (t+1)*g): V_t = sum_j (t*g + j + 1) * B[t*g + j].  A bucket that spans slices was summed
// by the fix-up into `fixed`; every other one is where the accumulation left it (`acc_buckets`).
// B counts the buckets of every polynomial of a batch (polynomial p owns [p * Bper, (p + 1) * Bper); g divides Bper, so a
// lane never straddles two polynomials) and the weight of a bucket is its index inside its polynomial, plus one.
template <class Cv, bool RAW>
__global__ __launch_bounds__(64) void k_msm_bucket_fold(const uint32_t* __restrict__ fixed,
                                                        const uint32_t* __restrict__ acc_buckets,
                                                        const uint32_t* __restrict__ off, SliceCfg T, uint32_t B,
                                                        uint32_t Bper, uint32_t g, uint32_t* __restrict__ out) {
    using G = Rep<Cv, RAW>;
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if ((uint64_t)t * g >= B) return;
    const uint32_t L = msm_slice_len(off[B], T);
    uint32_t lo = t * g, hi = (lo + g < B) ? lo + g : B;
    typename G::T run = G::identity(), sum = G::identity();
    for (uint32_t j = hi; j-- > lo;) {
        const uint32_t s = off[j], e = off[j + 1];
        const bool spans = e > s && s / L != (e - 1) / L;
        typename G::T q = G::load(spans ? fixed : acc_buckets, j);
        G::add(run, q);
        G::add(sum, run);
    }
    const uint32_t lo_in = lo & (Bper - 1u);
    if (lo_in) {
        typename G::T sh = rep_mul_small<Cv, RAW>(run, lo_in);
        G::add(sum, sh);
    }
    G::store(out, t, sum);
}

// out[block] = sum of in[block*256*K ...): K serial loads per lane, then an LDS tree.  RAW: elements in the 29-bit form;
// std_out: the block's sum is written as saturated XYZZ whatever the input form (the last launch of a tree)
// blockIdx.y: independent sums of a batch, m elements each, laid out back to back (input segment y starts at element
// y * m, its block sums at y * gridDim.x)
template <class Cv, bool RAW>
__global__ __launch_bounds__(256) void k_g1_sum(const uint32_t* __restrict__ in, uint32_t m, uint32_t per_lane, int std_out,
                                                uint32_t* __restrict__ out) {
    using G = Rep<Cv, RAW>;
    __shared__ uint32_t sh[256 * G::W];
    in += (size_t)blockIdx.y * m * G::W;
    out += (size_t)blockIdx.y * gridDim.x * (std_out ? 4 * Cv::Fq::N : G::W);
    uint32_t base = blockIdx.x * 256u * per_lane;
    typename G::T acc = G::identity();
    for (uint32_t k = 0; k < per_lane; k++) {
        uint32_t idx = base + k * 256u + threadIdx.x;
        if (idx < m) {
            typename G::T q = G::load(in, idx);
            G::add(acc, q);
        }
    }
    G::store(sh, threadIdx.x, acc);
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            typename G::T q = G::load(sh, threadIdx.x + s);
            G::add(acc, q);
            G::store(sh, threadIdx.x, acc);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (std_out) xyzz_store<Cv>(out + (size_t)blockIdx.x * (4 * Cv::Fq::N), G::to_std(acc));
        else G::store(out, blockIdx.x, acc);
    }
}

// XYZZ -> packed affine + u32 infinity flag
// one workgroup (its lane 0) per point: in_stride / out_stride in 32-bit words
template <class Cv>
__global__ void k_g1_to_affine(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t in_stride,
                               uint32_t out_stride) {
    constexpr int PT = 2 * Cv::Fq::N;
    if (threadIdx.x != 0) return;
    in += (size_t)blockIdx.x * in_stride;
    out += (size_t)blockIdx.x * out_stride;
    XYZZ<Cv> p = xyzz_load<Cv>(in);
    Affine<Cv> a = xyzz_to_affine<Cv>(p);
    affine_store<Cv>(out, a);
    out[PT] = p.is_inf() ? 1u : 0u;
}

template <class Cv>
__global__ void k_g1_set_identity(uint32_t* __restrict__ out) {   // one saturated XYZZ element per workgroup of 64
    constexpr int XZ = 4 * Cv::Fq::N;
    if (threadIdx.x < XZ) out[(size_t)blockIdx.x * XZ + threadIdx.x] = 0;
}

// ------------------------------------------------------------------------------------------
// SRS: packing, window-multiple table, test-SRS generation
// ------------------------------------------------------------------------------------------

// strided host records (x, y[, infinity byte]) -> packed affine, identity = zeros
template <class Cv>
__global__ __launch_bounds__(256) void k_srs_pack(const uint8_t* __restrict__ raw, uint32_t n, uint32_t stride,
                                                  uint32_t* __restrict__ out) {
    constexpr int PT = 2 * Cv::Fq::N;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* rec = raw + (size_t)i * stride;
    bool inf = (stride > PT * 4u) && rec[PT * 4] != 0;
    for (int k = 0; k < PT; k++) {
        uint32_t v;
        memcpy(&v, rec + 4 * k, 4);
        out[(size_t)i * PT + k] = inf ? 0u : v;
    }
}

template <class Q>
struct Jac {
    Fp<Q> X, Y, Z;
};
// Jacobian doubling, a = 0 (EFD dbl-2009-l)
template <class Q>
JF_DEV Jac<Q> jac_double(const Jac<Q>& p) {
    Jac<Q> r;
    auto A = fp_sqr<Q>(p.X);
    auto Bv = fp_sqr<Q>(p.Y);
    auto Cc = fp_sqr<Q>(Bv);
    auto t = fp_add<Q>(p.X, Bv);
    auto D = fp_dbl<Q>(fp_sub<Q>(fp_sub<Q>(fp_sqr<Q>(t), A), Cc));
    auto E = fp_add<Q>(fp_dbl<Q>(A), A);
    auto F = fp_sqr<Q>(E);
    r.X = fp_sub<Q>(F, fp_dbl<Q>(D));
    auto C8 = fp_dbl<Q>(fp_dbl<Q>(fp_dbl<Q>(Cc)));
    r.Y = fp_sub<Q>(fp_mul<Q>(E, fp_sub<Q>(D, r.X)), C8);
    r.Z = fp_dbl<Q>(fp_mul<Q>(p.Y, p.Z));
    return r;
}

// table[w][i] = 2^(c*w) * P_i for points [i0, i0 + m).  One lane per point: Jacobian doubling
// chain, then Montgomery's trick over the lane's own W-1 denominators (zs / pre are scratch).
template <class Cv>
__global__ __launch_bounds__(128) void k_srs_table(uint32_t* __restrict__ table, uint32_t n, uint32_t i0, uint32_t m,
                                                   int c, int W, uint32_t* __restrict__ zs,
                                                   uint32_t* __restrict__ pre) {
    using Q = typename Cv::Fq;
    constexpr int NQ = Q::N, PT = 2 * Q::N;
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    uint32_t i = i0 + k;
    Affine<Cv> p = affine_load<Cv>(table + (size_t)i * PT);
    if (p.is_inf()) {
        for (int w = 1; w < W; w++) affine_store<Cv>(table + ((size_t)w * n + i) * PT, p);
        return;
    }
    Jac<Q> J;
    J.X = p.x;
    J.Y = p.y;
    J.Z = Fp<Q>::one();
    Fp<Q> prod = Fp<Q>::one();
    for (int w = 1; w < W; w++) {
        for (int d = 0; d < c; d++) J = jac_double<Q>(J);
        bool inf = J.Z.is_zero();
        Fp<Q> zeff = inf ? Fp<Q>::one() : J.Z;
        Affine<Cv> tmp;
        tmp.x = inf ? Fp<Q>::zero() : J.X;
        tmp.y = inf ? Fp<Q>::zero() : J.Y;
        affine_store<Cv>(table + ((size_t)w * n + i) * PT, tmp);
        fp_store<Q>(zs + ((size_t)(w - 1) * m + k) * NQ, zeff);
        fp_store<Q>(pre + ((size_t)(w - 1) * m + k) * NQ, prod);
        prod = fp_mul<Q>(prod, zeff);
    }
    Fp<Q> inv = fp_inv_safegcd<Q>(prod);
    for (int w = W - 1; w >= 1; w--) {
        Fp<Q> z = fp_load<Q>(zs + ((size_t)(w - 1) * m + k) * NQ);
        Fp<Q> pr = fp_load<Q>(pre + ((size_t)(w - 1) * m + k) * NQ);
        Fp<Q> zi = fp_mul<Q>(inv, pr);
        inv = fp_mul<Q>(inv, z);
        Fp<Q> zi2 = fp_sqr<Q>(zi);
        uint32_t* slot = table + ((size_t)w * n + i) * PT;
        Affine<Cv> a = affine_load<Cv>(slot);
        a.x = fp_mul<Q>(a.x, zi2);
        a.y = fp_mul<Q>(a.y, fp_mul<Q>(zi2, zi));
        affine_store<Cv>(slot, a);
    }
}

// fixed-base table for test-SRS generation: fb[w][d] = d * 2^(8w) * g, w < 32, d < 256
template <class Cv>
__global__ __launch_bounds__(64) void k_fixed_base_table(const uint32_t* __restrict__ g, uint32_t* __restrict__ fb) {
    using Q = typename Cv::Fq;
    constexpr int PT = 2 * Q::N;
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 32 * 256) return;
    uint32_t w = id >> 8, d = id & 255;
    Affine<Cv> base = affine_load<Cv>(g);
    XYZZ<Cv> acc = XYZZ<Cv>::identity();
    if (d != 0 && !base.is_inf()) {
        // acc = d * g, then 8w doublings
        XYZZ<Cv> bp = xyzz_from_affine<Cv>(base);
        acc = xyzz_mul_small<Cv>(bp, d);
        for (uint32_t k = 0; k < 8 * w; k++) acc = xyzz_double<Cv>(acc);
    }
    affine_store<Cv>(fb + (size_t)id * PT, xyzz_to_affine<Cv>(acc));
}

// out[i] = [beta^i] g  (gen_srs_for_testing, reference pcs/src/univariate_kzg/srs.rs:175-225)
template <class Cv>
__global__ __launch_bounds__(128) void k_srs_gen(const uint32_t* __restrict__ beta, const uint32_t* __restrict__ fb,
                                                 uint32_t n, uint32_t* __restrict__ out) {
    using Q = typename Cv::Fq;
    using Fr = typename Cv::Fr;
    constexpr int PT = 2 * Q::N;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<Fr> s = fp_from_mont<Fr>(fp_pow_u64<Fr>(fp_load<Fr>(beta), i));
    XYZZ<Cv> acc = XYZZ<Cv>::identity();
    for (int w = 0; w < 32; w++) {
        uint32_t d = extract_bits<Fr::N>(s.v, 8 * w, 8);
        if (d) {
            Affine<Cv> p = affine_load<Cv>(fb + (size_t)(w * 256 + d) * PT);
            xyzz_add_affine<Cv>(acc, p);
        }
    }
    affine_store<Cv>(out + (size_t)i * PT, xyzz_to_affine<Cv>(acc));
}

// ---- small group ops for the test surface ---------------------------------------------------
// ---- ark-serialize G1Affine records -> packed affine Montgomery points -------------------------------------
// limb i of (p + 1) / 4 (both base fields are 3 mod 4: a square root of a residue a is a^((p+1)/4))
template <class C>
JF_HD constexpr uint32_t sqrt_exp_limb(int i) {
    // (p + 1) >> 2 on the limbs, with the carry of the +1
    uint32_t carry = 1, lo = 0, hi = 0;
    for (int k = 0; k <= i + 1 && k < C::N; k++) {
        uint64_t t = (uint64_t)C::P(k) + carry;
        carry = (uint32_t)(t >> 32);
        if (k == i) lo = (uint32_t)t;
        if (k == i + 1) hi = (uint32_t)t;
    }
    if (i + 1 >= C::N) hi = carry;   // p + 1 never carries out of N limbs for these moduli, kept for exactness
    return (lo >> 2) | (hi << 30);
}
template <class C>
__device__ __noinline__ Fp<C> fp_sqrt_candidate(const Fp<C>& a) {
    constexpr int N = C::N;
    Fp<C> acc = Fp<C>::one();
    for (int i = N - 1; i >= 0; i--) {
        uint32_t e = 0;
#pragma unroll
        for (int j = 0; j < N; j++) e = (j == i) ? sqrt_exp_limb<C>(j) : e;
        for (int b = 31; b >= 0; b--) {
            acc = fp_sqr<C>(acc);
            if ((e >> b) & 1u) acc = fp_mul<C>(acc, a);
        }
    }
    return acc;
}
// a > b on canonical (non-Montgomery) limbs
template <class C>
JF_DEV bool fp_canon_gt(const Fp<C>& a, const Fp<C>& b) {
    for (int i = C::N - 1; i >= 0; i--) {
        if (a.v[i] != b.v[i]) return a.v[i] > b.v[i];
    }
    return false;
}
template <class C>
JF_DEV bool fp_canon_lt_p(const Fp<C>& a) {
    for (int i = C::N - 1; i >= 0; i--) {
        if (a.v[i] != C::P(i)) return a.v[i] < C::P(i);
    }
    return false;
}
// limb i of the scalar-field modulus r, as an exponent for the subgroup check
template <class Cv>
JF_DEV XYZZ<Cv> xyzz_mul_by_r(const Affine<Cv>& p) {
    using Fr = typename Cv::Fr;
    XYZZ<Cv> r = XYZZ<Cv>::identity();
    for (int i = Fr::N - 1; i >= 0; i--) {
        uint32_t e = 0;
#pragma unroll
        for (int j = 0; j < Fr::N; j++) e = (j == i) ? Fr::P(j) : e;
        for (int b = 31; b >= 0; b--) {
            r = xyzz_double<Cv>(r);
            if ((e >> b) & 1u) xyzz_add_affine<Cv>(r, p);
        }
    }
    return r;
}

// status[0]: OR of failure bits (1 not canonical, 2 not on the curve / no square root, 4 not in the r-torsion subgroup,
// 8 malformed flags), status[1]: smallest failing index (atomicMin).
// Formats (ark-serialize 0.4/0.5):
//   BN254 (ark-ec short Weierstrass default): x little-endian, flags in the top bits of the LAST byte (0x80: y is the
//     larger of {y, -y}; 0x40: infinity); uncompressed appends y little-endian, which carries the flags instead.
//   BLS12-381 (ark-bls12-381's own impl = the zcash encoding): x big-endian, flags in the top bits of the FIRST byte
//     (0x80 compressed, 0x40 infinity, 0x20 y is the larger); uncompressed appends y big-endian.
template <class Cv>
__global__ __launch_bounds__(128) void k_srs_decode(const uint8_t* __restrict__ raw, uint32_t n, int compressed,
                                                    int validate, uint32_t* __restrict__ out,
                                                    uint32_t* __restrict__ status) {
    using Q = typename Cv::Fq;
    constexpr int N = Q::N, FB = 4 * N, PT = 2 * N;
    constexpr bool BE = Cv::ID == 1;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t rec = compressed ? FB : 2 * FB;
    const uint8_t* src = raw + (size_t)i * rec;
    auto fail = [&](uint32_t bit) {
        atomicOr(&status[0], bit);
        atomicMin(&status[1], i);
        for (int k = 0; k < PT; k++) out[(size_t)i * PT + k] = 0;
    };
    auto read = [&](const uint8_t* b, Fp<Q>& f) {
        for (int w = 0; w < N; w++) {
            uint32_t v = 0;
            for (int k = 0; k < 4; k++) {
                const int byte = 4 * w + k;   // little-endian byte index
                v |= (uint32_t)b[BE ? FB - 1 - byte : byte] << (8 * k);
            }
            f.v[w] = v;
        }
    };
    // flags live in the most significant byte of x (compressed) or of the flag-carrying coordinate
    const uint8_t* fsrc = BE ? src : (compressed ? src : src + FB);
    const uint8_t fbyte = BE ? fsrc[0] : fsrc[FB - 1];
    bool infinity, y_larger;
    if (BE) {
        const bool cflag = (fbyte & 0x80) != 0;
        infinity = (fbyte & 0x40) != 0;
        y_larger = (fbyte & 0x20) != 0;
        if (cflag != (compressed != 0) || (!compressed && y_larger)) return fail(8);
    } else {
        infinity = (fbyte & 0x40) != 0;
        y_larger = (fbyte & 0x80) != 0;
        if (infinity && y_larger) return fail(8);
    }
    Fp<Q> x, y;
    read(src, x);
    if (!compressed) read(src + FB, y);
    // clear the flag bits from the coordinate that carried them
    const uint32_t clear = BE ? 0x1fffffffu : 0x3fffffffu;
    if (BE || compressed) x.v[N - 1] &= clear;
    else y.v[N - 1] &= clear;
    if (infinity) {
        bool zero = x.is_zero() && (compressed || y.is_zero());
        if (!zero) return fail(8);
        for (int k = 0; k < PT; k++) out[(size_t)i * PT + k] = 0;
        return;
    }
    if (!fp_canon_lt_p<Q>(x) || (!compressed && !fp_canon_lt_p<Q>(y))) return fail(1);
    Affine<Cv> p;
    p.x = fp_to_mont<Q>(x);
    // the curve coefficient b (3 for BN254, 4 for BLS12-381) in Montgomery form, by additions of one()
    const Fp<Q> two = fp_dbl<Q>(Fp<Q>::one());
    const Fp<Q> bcoef = Cv::ID == 0 ? fp_add<Q>(two, Fp<Q>::one()) : fp_dbl<Q>(two);
    const Fp<Q> rhs = fp_add<Q>(fp_mul<Q>(fp_sqr<Q>(p.x), p.x), bcoef);
    if (compressed) {
        Fp<Q> cand = fp_sqrt_candidate<Q>(rhs);
        if (!fp_eq<Q>(fp_sqr<Q>(cand), rhs)) return fail(2);
        const Fp<Q> neg = fp_neg<Q>(cand);
        const bool cand_larger = fp_canon_gt<Q>(fp_from_mont<Q>(cand), fp_from_mont<Q>(neg));
        p.y = (cand_larger == y_larger) ? cand : neg;
    } else {
        p.y = fp_to_mont<Q>(y);
        if (validate && !fp_eq<Q>(fp_sqr<Q>(p.y), rhs)) return fail(2);
    }
    if (validate && Cv::ID == 1) {   // BN254 G1 has cofactor 1
        XYZZ<Cv> t = xyzz_mul_by_r<Cv>(p);
        if (!t.is_inf()) return fail(4);
    }
    affine_store<Cv>(out + (size_t)i * PT, p);
}

int32_t srs_decode_points(int32_t curve, const uint8_t* d_raw, size_t n, int compressed, int validate, uint32_t* d_out,
                          uint32_t* d_status, jfhip_stream* st) {
    dim3 grid((unsigned)((n + 127) / 128));
    if (curve == JFHIP_BN254)
        hipLaunchKernelGGL((k_srs_decode<Bn254G1>), grid, dim3(128), 0, st->s, d_raw, (uint32_t)n, compressed, validate,
                           d_out, d_status);
    else
        hipLaunchKernelGGL((k_srs_decode<Bls381G1>), grid, dim3(128), 0, st->s, d_raw, (uint32_t)n, compressed, validate,
                           d_out, d_status);
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

template <class Cv>
__global__ void k_g1_op(int op, const uint32_t* __restrict__ P, const uint32_t* __restrict__ Qp,
                        uint32_t* __restrict__ out) {
    using Fr = typename Cv::Fr;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Affine<Cv> p = affine_load<Cv>(P);
    XYZZ<Cv> acc = xyzz_from_affine<Cv>(p);
    if (op == 0) {
        Affine<Cv> q = affine_load<Cv>(Qp);
        xyzz_add_affine<Cv>(acc, q);
    } else if (op == 4) {
        // the group-element DFT's scalar multiplication (g29.hpp p29_mul_scalar: non-adjacent form on 29-bit limbs)
        const P29<Cv> r = p29_mul_scalar<Cv>(p29_from_std<Cv>(acc), fp_from_mont<Fr>(fp_load<Fr>(Qp)));
        acc = xyzz29_to_std<Cv>(r.v, r.inf);
    } else if (op == 2 || op == 3) {
        // the MSM tail's group law on lazily reduced 29-bit limbs (p29_add / p29_double), same results as ops 0 / 1:
        // op 3's 256-step double-and-add chain feeds every output back in, which is what the stated bounds must survive
        P29<Cv> a = p29_from_std<Cv>(acc);
        if (op == 2) {
            P29<Cv> b = p29_from_std<Cv>(xyzz_from_affine<Cv>(affine_load<Cv>(Qp)));
            p29_add<Cv>(a, b);
            acc = xyzz29_to_std<Cv>(a.v, a.inf);
        } else {
            Fp<Fr> k = fp_from_mont<Fr>(fp_load<Fr>(Qp));
            P29<Cv> r = Rep<Cv, true>::identity();
            for (int bit = 32 * Fr::N - 1; bit >= 0; bit--) {
                r = p29_double<Cv>(r);
                if (extract_bits<Fr::N>(k.v, bit, 1)) p29_add<Cv>(r, a);
            }
            acc = xyzz29_to_std<Cv>(r.v, r.inf);
        }
    } else {
        Fp<Fr> k = fp_from_mont<Fr>(fp_load<Fr>(Qp));
        XYZZ<Cv> r = XYZZ<Cv>::identity();
        for (int bit = 32 * Fr::N - 1; bit >= 0; bit--) {
            r = xyzz_double<Cv>(r);
            if (extract_bits<Fr::N>(k.v, bit, 1)) xyzz_add<Cv>(r, acc);
        }
        acc = r;
    }
    affine_store<Cv>(out, xyzz_to_affine<Cv>(acc));
}

// ------------------------------------------------------------------------------------------
// host drivers
// ------------------------------------------------------------------------------------------

int windows_for(int32_t curve, int c) {
    int bits = (curve == JFHIP_BN254 ? 254 : 255) + 1;  // +1: signed-digit carry
    return (bits + c - 1) / c;
}

int choose_window_bits(int32_t curve, size_t n) {
    // minimise n * W(c) * 10 + 2^(c-1) * 60 field multiplications (accumulate + fold): reproduces the measured optimum
    // for BN254 (scripts/window_sweep.py: c = 17 at 2^20, 22 at 2^24)
    int best = 8;
    double best_cost = 1e300;
    for (int c = 6; c <= 23; c++) {
        int W = windows_for(curve, c);
        if ((double)W * (double)n >= 2147483648.0) continue;
        double cost = (double)n * W * 10.0 + (double)(1u << (c - 1)) * 60.0;
        if (cost < best_cost) {
            best_cost = cost;
            best = c;
        }
    }
    // BN254 around 2^21 points (the per-GPU shard of the metric's MSM on 8 GPUs): the model picks c = 20 (13 windows, three
    // sort levels, 2^19 buckets to fold); measured (scripts/window_sweep.py 21) c = 15 3.58 ms, 16 3.29, 17 3.05, 18 3.36,
    // 19 3.57, 20 3.29-3.32, 21 3.59: two sort levels and a 1.9 GiB table (inside the translation reach) win
    if (curve == JFHIP_BN254 && n > ((size_t)3 << 19) && n <= ((size_t)5 << 19)) best = 17;
    // BLS12-381: the 12-limb fold / fix-up tails weigh more than the model says around 2^20 points; measured
    // (scripts/msm_probe_bls.py) 2^20: c = 16 5.3 ms, 18 5.5, 19 6.1; 2^21: c = 20 8.4 ms, 19 9.6, 18 9.6; 2^22: c = 20
    if (curve != JFHIP_BN254) {
        if (n <= ((size_t)3 << 19)) best = best < 16 ? best : 16;
        else if (n <= ((size_t)3 << 21)) best = 20;
    }
    return best;
}

template <class Cv>
static int32_t srs_build_table_t(jfhip_srs* srs, jfhip_stream* st) {
    constexpr int NQ = Cv::Fq::N;
    const size_t n = srs->n;
    if (srs->W <= 1) return JFHIP_OK;
    const size_t tile = n < ((size_t)1 << 20) ? n : ((size_t)1 << 20);
    DevBuf zs, pre;
    JF_TRY(zs.ensure((size_t)(srs->W - 1) * tile * NQ * 4));
    JF_TRY(pre.ensure((size_t)(srs->W - 1) * tile * NQ * 4));
    for (size_t i0 = 0; i0 < n; i0 += tile) {
        uint32_t m = (uint32_t)((n - i0 < tile) ? n - i0 : tile);
        hipLaunchKernelGGL((k_srs_table<Cv>), dim3((m + 127) / 128), dim3(128), 0, st->s, srs->table, (uint32_t)n,
                           (uint32_t)i0, m, srs->c, srs->W, zs.as<uint32_t>(), pre.as<uint32_t>());
    }
    JF_HIP(hipGetLastError());
    JF_HIP(hipStreamSynchronize(st->s));
    zs.release();
    pre.release();
    return JFHIP_OK;
}

// Which arithmetic the bucket accumulation of an SRS uses.  Measured on MI355X (uniform scalars), bucket_accumulate phase,
// 29-bit limbs vs saturated: BN254 2^20 points 1.24 vs 1.59 ms, 2^22 4.24 vs 5.29, 2^24 14.9 vs 18.3 -- the 29-bit kernel
// everywhere (profiles/r02_*; before the gather of its software pipeline was moved behind the bucket hand-over it LOST at
// 2^24, 20.7 ms, and BLS12-381 at every size).  BLS12-381 (14 limbs of 29 bits): 2^20 2.97 vs 3.56 ms, 2^22 8.99 vs 11.28.
// JFHIP_MSM_F29=0 / 1 forces saturated / 29-bit limbs (A/B measurements).
static bool use_f29(int32_t curve, size_t n) {
    const char* e = getenv("JFHIP_MSM_F29");
    if (e && e[0] == '0') return false;
    if (e && e[0] == '1') return true;
    (void)curve;
    (void)n;
    return true;
}

int32_t srs_build_table(jfhip_srs* srs, const uint32_t* d_points_packed, jfhip_stream* st) {
    const int PT = 2 * fq_limbs32(srs->curve);
    if (d_points_packed != srs->table)
        JF_HIP(hipMemcpyAsync(srs->table, d_points_packed, srs->n * PT * 4, hipMemcpyDeviceToDevice, st->s));
    int32_t rc = (srs->curve == JFHIP_BN254) ? srs_build_table_t<Bn254G1>(srs, st) : srs_build_table_t<Bls381G1>(srs, st);
    if (rc != JFHIP_OK) return rc;
    if (use_f29(srs->curve, srs->n)) {
        // the whole table moves to the Montgomery factor of the 29-bit-limb arithmetic (fp29.hpp)
        JF_TRY(table_rescale(srs->curve, srs->table, (size_t)(srs->W > 0 ? srs->W : 1) * srs->n, 1, st->s));
        JF_HIP(hipStreamSynchronize(st->s));
        srs->f29 = true;
    }
    return JFHIP_OK;
}

int32_t srs_pack_points(int32_t curve, const uint8_t* d_raw, size_t n, size_t stride, uint32_t* d_out,
                        jfhip_stream* st) {
    dim3 grid((unsigned)((n + 255) / 256));
    if (curve == JFHIP_BN254)
        hipLaunchKernelGGL((k_srs_pack<Bn254G1>), grid, dim3(256), 0, st->s, d_raw, (uint32_t)n, (uint32_t)stride,
                           d_out);
    else
        hipLaunchKernelGGL((k_srs_pack<Bls381G1>), grid, dim3(256), 0, st->s, d_raw, (uint32_t)n, (uint32_t)stride,
                           d_out);
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

template <class Cv>
static int32_t srs_gen_t(const uint64_t* beta_mont, const uint64_t* g_xy, size_t n, uint32_t* d_out,
                         jfhip_stream* st) {
    constexpr int PT = 2 * Cv::Fq::N;
    DevBuf par, fb;
    JF_TRY(par.ensure((8 + PT) * 4));
    JF_TRY(fb.ensure((size_t)32 * 256 * PT * 4));
    JF_HIP(hipMemcpyAsync(par.p, beta_mont, 32, hipMemcpyHostToDevice, st->s));
    JF_HIP(hipMemcpyAsync(par.as<uint32_t>() + 8, g_xy, PT * 4, hipMemcpyHostToDevice, st->s));
    hipLaunchKernelGGL((k_fixed_base_table<Cv>), dim3(32 * 256 / 64), dim3(64), 0, st->s, par.as<uint32_t>() + 8,
                       fb.as<uint32_t>());
    hipLaunchKernelGGL((k_srs_gen<Cv>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st->s, par.as<uint32_t>(),
                       fb.as<uint32_t>(), (uint32_t)n, d_out);
    JF_HIP(hipGetLastError());
    JF_HIP(hipStreamSynchronize(st->s));
    par.release();
    fb.release();
    return JFHIP_OK;
}

int32_t srs_gen_points(int32_t curve, const uint64_t* beta_mont, const uint64_t* g_xy, size_t n, uint32_t* d_out,
                       jfhip_stream* st) {
    if (curve == JFHIP_BN254) return srs_gen_t<Bn254G1>(beta_mont, g_xy, n, d_out, st);
    return srs_gen_t<Bls381G1>(beta_mont, g_xy, n, d_out, st);
}

// tree of k_g1_sum launches, ping-ponging between red0 / red1; the final points -> d_out_xyzz, always saturated XYZZ.
// RAW: d_in holds elements in the 29-bit form (Rep<Cv, true>).  segs independent sums of `count` elements each, back to
// back in d_in; their results back to back in d_out_xyzz.
template <class Cv, bool RAW>
static int32_t g1_sum_xyzz_t(const uint32_t* d_in, size_t count, uint32_t* d_out_xyzz, jfhip_stream* st, size_t segs = 1) {
    constexpr int XW = Rep<Cv, RAW>::W;
    if (count == 0) {
        hipLaunchKernelGGL((k_g1_set_identity<Cv>), dim3((unsigned)segs), dim3(64), 0, st->s, d_out_xyzz);
        JF_HIP(hipGetLastError());
        return JFHIP_OK;
    }
    const uint32_t* cur = d_in;
    size_t m = count;
    int flip = 0;
    for (;;) {
        uint32_t per_lane = m > 65536 ? 4 : 1;
        size_t blocks = (m + 256 * per_lane - 1) / (256 * per_lane);
        uint32_t* dst;
        if (blocks == 1) dst = d_out_xyzz;
        else {
            DevBuf& buf = flip ? st->red1 : st->red0;
            JF_TRY(buf.ensure(segs * blocks * XW * 4));
            dst = buf.as<uint32_t>();
        }
        hipLaunchKernelGGL((k_g1_sum<Cv, RAW>), dim3((unsigned)blocks, (unsigned)segs), dim3(256), 0, st->s, cur,
                           (uint32_t)m, per_lane, blocks == 1 ? 1 : 0, dst);
        JF_HIP(hipGetLastError());
        if (blocks == 1) break;
        cur = dst;
        m = blocks;
        flip ^= 1;
    }
    return JFHIP_OK;
}

// The MSMs of a batch against one window table in ONE pipeline (UnivariateKzgPCS::batch_commit, reference
// pcs/src/univariate_kzg/mod.rs:128-140; a single commit is the batch of one): the bucket id of polynomial p is widened
// to p * 2^(c-1) + bucket, so one sort, one accumulation, one fix-up, one fold and one sum tree serve every polynomial
// and the latency-bound tail (fix-up, fold, tree: ~0.45 ms of a 2^20-point MSM) is paid once per commit round instead of
// once per polynomial.  d_out_xyzz: mb.count saturated XYZZ elements.
template <class Cv>
static int32_t msm_multi_t(const jfhip_srs* srs, MsmBatch mb, uint32_t* d_out_xyzz, jfhip_stream* st) {
    constexpr int XZ = 4 * Cv::Fq::N;
    PhaseTimer pt(st, 0);
    const uint32_t P = mb.count;
    const int c = srs->c, W = srs->W;
    const uint32_t Bper = 1u << (c - 1);
    static const int env_tile = getenv("JFHIP_MSM_TILE") ? atoi(getenv("JFHIP_MSM_TILE")) : 0;
    const uint32_t tile_scalars = env_tile >= 256 ? (uint32_t)(env_tile / 256 * 256) : (uint32_t)SORT_TILE_SCALARS;
    size_t n_sum = 0;
    mb.tile0[0] = 0;
    for (uint32_t p = 0; p < P; p++) {
        n_sum += mb.n[p];
        mb.tile0[p + 1] = mb.tile0[p] + (uint32_t)((mb.n[p] + tile_scalars - 1) / tile_scalars);
    }
    if (n_sum == 0) {
        hipLaunchKernelGGL((k_g1_set_identity<Cv>), dim3(P), dim3(64), 0, st->s, d_out_xyzz);
        JF_HIP(hipGetLastError());
        pt.finish();
        return JFHIP_OK;
    }
    const size_t Bsz = (size_t)Bper * P;
    const size_t total_max = n_sum * (size_t)W;
    if (total_max >= ((size_t)1 << 32) || Bsz >= ((size_t)1 << 31)) {
        set_error("msm too large: n * windows = %zu exceeds 2^32", total_max);
        return JFHIP_ERR_INVALID;
    }
    const uint32_t B = (uint32_t)Bsz;
    // slice length: one round of resident lanes (256 CUs x 4 SIMDs x 64 lanes x 4 waves of the saturated kernel, 3 of the
    // 29-bit one: a second, partly filled round runs at a fraction of the chip), at most 256 entries per lane
    static const int f29_occ = [] { const char* e = getenv("JFHIP_MSM_F29_OCC"); return e ? atoi(e) : 3; }();
    const size_t lanes = (size_t)65536 * (srs->f29 ? (f29_occ >= 2 && f29_occ <= 4 ? f29_occ : 3) : 4);
    // (more entries than lanes x 256: a whole number of such rounds)
    const size_t rounds = (total_max + lanes * 256 - 1) / (lanes * 256);
    size_t L = (total_max + lanes * rounds - 1) / (lanes * rounds);
    if (L < 16) L = 16;
    const size_t T = (total_max + L - 1) / L;
    // the kernels derive the slice length from off[B]; JFHIP_MSM_FIXED_SLICE=1 pins it to the host's L (rounds 1-2, A/B)
    static const bool fixed_slice = getenv("JFHIP_MSM_FIXED_SLICE") && getenv("JFHIP_MSM_FIXED_SLICE")[0] == '1';
    static const int env_min_slice = getenv("JFHIP_MSM_MIN_SLICE") ? atoi(getenv("JFHIP_MSM_MIN_SLICE")) : 0;
    const size_t min_slice = env_min_slice >= 16 && env_min_slice <= 256 ? (size_t)env_min_slice : 16;   // 16 / 32 / 64 measured
    const SliceCfg scfg{(uint32_t)T, (uint32_t)(fixed_slice || L < min_slice ? L : min_slice)};
    DigitPlan pl;
    pl.c = c;
    pl.W = W;
    // bucket id bits are consumed in levels of <= 9 bits (staged scatters: runs of >= 16 entries per key and chunk); the
    // polynomial index of a batch rides on top of the coarse level's bin id and costs no level
    static const bool staged = !(getenv("JFHIP_MSM_SORT") && getenv("JFHIP_MSM_SORT")[0] == '0');
    static const int env_cb = getenv("JFHIP_MSM_COARSE_BITS") ? atoi(getenv("JFHIP_MSM_COARSE_BITS")) : -1;
    const int level_max = staged ? 9 : 8;
    if (staged) {
        const int nlev = (c - 1 + 8) / 9;
        pl.coarse_bits = (c - 1 + nlev - 1) / nlev;
    } else {
        pl.coarse_bits = (c - 1) - 8;
        if (pl.coarse_bits > MAX_COARSE_BITS) pl.coarse_bits = MAX_COARSE_BITS;
    }
    if (env_cb >= 0 && env_cb < pl.coarse_bits) pl.coarse_bits = env_cb;
    if (pl.coarse_bits < 0) pl.coarse_bits = 0;
    pl.fine_bits = (c - 1) - pl.coarse_bits;
    if (pl.fine_bits > MAX_FINE_BITS) {
        set_error("window_bits %d unsupported by the bucket sort (max %d)", c, MAX_COARSE_BITS + MAX_FINE_BITS + 1);
        return JFHIP_ERR_INVALID;
    }
    // a flush stages the <= 16 digits of 256 scalars; more windows than that (c < 16, small MSMs): direct scatter
    const bool staged_coarse = staged && W <= 16;
    const uint32_t nb = 1u << pl.coarse_bits;    // coarse bins of one polynomial
    const uint32_t nbg = nb * P;                 // ... of the batch

    JF_TRY(st->dig.ensure(total_max * 8));     // level-1 entries (payload, bucket)
    JF_TRY(st->sorted.ensure(total_max * 4));
    JF_TRY(st->counts.ensure(((size_t)nbg + 1) * 4));
    JF_TRY(st->offs.ensure(((size_t)B + 1) * 4));
    JF_TRY(st->cursor.ensure(((size_t)nbg + 1) * 4));
    JF_TRY(st->bsum.ensure(((size_t)nbg + 1) * 4));
    constexpr size_t XW = 4 * F29<typename Cv::Fq>::L;   // words of a raw (29-bit form) element, >= XZ
    JF_TRY(st->buckets.ensure((size_t)B * XW * 4));
    // fold: g buckets per lane, then a sum tree (the fold output reuses partA after the fix-up)
    // latency-bound for small bucket counts (few lanes, deep chains): short chains there
    static const int env_g = getenv("JFHIP_MSM_FOLD_G") ? atoi(getenv("JFHIP_MSM_FOLD_G")) : 0;
    uint32_t g = (Bper <= (1u << 17)) ? 4 : (Bper <= (1u << 20)) ? 8 : 16;   // measured: BLS12-381 2^22 points (B = 2^19) 0.91 -> 0.73 ms with 8
    if (env_g > 0 && (env_g & (env_g - 1)) == 0 && (uint32_t)env_g <= Bper) g = (uint32_t)env_g;
    if (g > Bper) g = Bper;
    const uint32_t Tf_per = Bper / g;            // fold outputs of one polynomial (g divides Bper: powers of two)
    const size_t Tf = (size_t)Tf_per * P;
    JF_TRY(st->partA.ensure((T > Tf ? T : Tf) * XW * 4));
    JF_TRY(st->partB.ensure(T * XZ * 4));

    uint32_t* counts = st->counts.as<uint32_t>();
    uint32_t* offs = st->offs.as<uint32_t>();
    uint32_t* cursor = st->cursor.as<uint32_t>();
    uint32_t* cbase = st->bsum.as<uint32_t>();
    uint32_t* buckets = st->buckets.as<uint32_t>();
    uint2* entries = st->dig.as<uint2>();

    auto scan = [&](uint32_t* cnts, uint32_t m, uint32_t* out_base, uint32_t* out_cursor) -> int32_t {
        // out_base[0..m] = exclusive scan (out_base[m] = total), out_cursor (optional) = copy of out_base[0..m)
        const uint32_t nblk = (m + SCAN_BLOCK * SCAN_ITEMS - 1) / (SCAN_BLOCK * SCAN_ITEMS);
        JF_TRY(st->red0.ensure((size_t)nblk * 4));
        uint32_t* bs = st->red0.as<uint32_t>();
        hipLaunchKernelGGL(k_scan_local, dim3(nblk), dim3(SCAN_BLOCK), 0, st->s, cnts, m, out_base, bs);
        hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SCAN_BLOCK), 0, st->s, bs, nblk, out_base + m);
        hipLaunchKernelGGL(k_scan_add, dim3((m + 255) / 256), dim3(256), 0, st->s, out_base, m, bs, out_cursor);
        return JFHIP_OK;
    };
    const unsigned tiles = mb.tile0[P];
    if (staged_coarse) {
        // per-tile histograms in [polynomial][bin][tile] order; their exclusive scan places every tile in every coarse bin
        const size_t m = (size_t)nb * tiles;
        JF_TRY(st->tilehist.ensure((m + 1) * 4));
        JF_TRY(st->tilebase.ensure((m + 1) * 4));
        hipLaunchKernelGGL((k_msm_coarse<Cv, 2>), dim3(tiles), dim3(256), 0, st->s, mb, pl, tile_scalars,
                           (uint32_t)srs->n, st->tilehist.as<uint32_t>(), entries);
        JF_TRY(scan(st->tilehist.as<uint32_t>(), (uint32_t)m, st->tilebase.as<uint32_t>(), (uint32_t*)nullptr));
        hipLaunchKernelGGL(k_msm_coarse_bases, dim3((nbg + 256) / 256), dim3(256), 0, st->s, st->tilebase.as<uint32_t>(), nb,
                           mb, cbase, offs + B);
        pt.mark();
        hipLaunchKernelGGL((k_msm_coarse_staged<Cv>), dim3(tiles), dim3(256), 0, st->s, mb, pl, tile_scalars,
                           (uint32_t)srs->n, st->tilebase.as<uint32_t>(), entries);
    } else {
        JF_HIP(hipMemsetAsync(counts, 0, ((size_t)nbg + 1) * 4, st->s));
        hipLaunchKernelGGL((k_msm_coarse<Cv, 0>), dim3(tiles), dim3(256), 0, st->s, mb, pl, tile_scalars,
                           (uint32_t)srs->n, counts, entries);
        JF_TRY(scan(counts, nbg, cbase, cursor));
        JF_HIP(hipMemcpyAsync(offs + B, cbase + nbg, 4, hipMemcpyDeviceToDevice, st->s));
        pt.mark();
        hipLaunchKernelGGL((k_msm_coarse<Cv, 1>), dim3(tiles), dim3(256), 0, st->s, mb, pl, tile_scalars,
                           (uint32_t)srs->n, cursor, entries);
    }
    {
        // refinement levels over the fine bits: one level if it fits, else two
        const unsigned chunks = (unsigned)((total_max + LVL_CHUNK - 1) / LVL_CHUNK);
        auto scatter = [&](bool final, const uint2* in, const uint32_t* segbase, uint32_t nseg, int shift, int bits,
                           uint32_t* cur, uint2* out_e, uint32_t* out_s) {
            const dim3 grid(staged ? (unsigned)((total_max + STG_CAP - 1) / STG_CAP) : chunks), block(LVL_THREADS);
            if (staged && final)
                hipLaunchKernelGGL((k_msm_level_staged<2>), grid, block, 0, st->s, in, segbase, nseg, shift, bits, cur, out_e, out_s);
            else if (staged)
                hipLaunchKernelGGL((k_msm_level_staged<1>), grid, block, 0, st->s, in, segbase, nseg, shift, bits, cur, out_e, out_s);
            else if (final)
                hipLaunchKernelGGL((k_msm_level<2>), grid, block, 0, st->s, in, segbase, nseg, shift, bits, cur, out_e, out_s);
            else
                hipLaunchKernelGGL((k_msm_level<1>), grid, block, 0, st->s, in, segbase, nseg, shift, bits, cur, out_e, out_s);
        };
        const uint2* cur_entries = entries;
        const uint32_t* cur_base = cbase;
        uint32_t cur_nseg = nbg;
        int rem = pl.fine_bits;
        if (rem > level_max) {
            const int lo_bits = rem / 2, hi_bits = rem - lo_bits;
            const uint32_t nmid = nbg << hi_bits;
            JF_TRY(st->midbase.ensure(((size_t)nmid + 1) * 4));
            JF_TRY(st->midcnt.ensure(((size_t)nmid + 1) * 4));
            JF_TRY(st->midcur.ensure(((size_t)nmid + 1) * 4));
            JF_TRY(st->dig2.ensure(total_max * 8));
            JF_HIP(hipMemsetAsync(st->midcnt.p, 0, ((size_t)nmid + 1) * 4, st->s));
            hipLaunchKernelGGL((k_msm_level<0>), dim3(chunks), dim3(LVL_THREADS), 0, st->s, cur_entries, cur_base,
                               cur_nseg, lo_bits, hi_bits, st->midcnt.as<uint32_t>(), (uint2*)nullptr,
                               (uint32_t*)nullptr);
            JF_TRY(scan(st->midcnt.as<uint32_t>(), nmid, st->midbase.as<uint32_t>(), st->midcur.as<uint32_t>()));
            scatter(false, cur_entries, cur_base, cur_nseg, lo_bits, hi_bits, st->midcur.as<uint32_t>(),
                    st->dig2.as<uint2>(), (uint32_t*)nullptr);
            cur_entries = st->dig2.as<uint2>();
            cur_base = st->midbase.as<uint32_t>();
            cur_nseg = nmid;
            rem = lo_bits;
        }
        JF_TRY(st->finecnt.ensure(((size_t)B + 1) * 4));
        JF_TRY(st->finecur.ensure(((size_t)B + 1) * 4));
        JF_HIP(hipMemsetAsync(st->finecnt.p, 0, ((size_t)B + 1) * 4, st->s));
        hipLaunchKernelGGL((k_msm_level<0>), dim3(chunks), dim3(LVL_THREADS), 0, st->s, cur_entries, cur_base,
                           cur_nseg, 0, rem, st->finecnt.as<uint32_t>(), (uint2*)nullptr, (uint32_t*)nullptr);
        JF_TRY(scan(st->finecnt.as<uint32_t>(), B, offs, st->finecur.as<uint32_t>()));
        scatter(true, cur_entries, cur_base, cur_nseg, 0, rem, st->finecur.as<uint32_t>(), (uint2*)nullptr,
                st->sorted.as<uint32_t>());
    }
    pt.mark();
    JF_TRY(st->heavy.ensure((T / HEAVY_SPAN + 4) * 4));
    JF_HIP(hipMemsetAsync(st->heavy.p, 0, 4, st->s));
    uint32_t* fold = st->partA.as<uint32_t>();
    if (srs->f29) {
        const int occ = f29_occ;
        auto kern = occ == 2 ? k_msm_accumulate29<Cv, 2> : occ == 3 ? k_msm_accumulate29<Cv, 3> : k_msm_accumulate29<Cv, 4>;
        constexpr size_t XR = 4 * F29<typename Cv::Fq>::L;
        const size_t elems = (size_t)B + 2 * T;
        JF_TRY(st->raw29.ensure(elems * XR * 4));
        uint32_t* raw = st->raw29.as<uint32_t>();
        uint32_t* rawA = raw + (size_t)B * XR;
        uint32_t* rawB = raw + ((size_t)B + T) * XR;
        JF_HIP(hipMemsetAsync(raw, 0, (size_t)B * XR * 4, st->s));  // untouched buckets are the identity
        hipLaunchKernelGGL(kern, dim3((unsigned)((T + 127) / 128)), dim3(128), 0, st->s, st->sorted.as<uint32_t>(), offs,
                           B, srs->table, scfg, raw, rawA, rawB);
        pt.mark();
        // the runs stay in the accumulation's raw form through the fix-up, the fold and the sum tree
        hipLaunchKernelGGL((k_msm_fixup<Cv, true>), dim3((B + 127) / 128), dim3(128), 0, st->s, offs, B, scfg, rawA,
                           rawB, buckets, st->heavy.as<uint32_t>());
        hipLaunchKernelGGL((k_msm_fixup_heavy_chunks<Cv, true>), dim3(1024), dim3(256), 0, st->s, offs, B, scfg, rawA,
                           st->heavy.as<uint32_t>());
        hipLaunchKernelGGL((k_msm_fixup_heavy<Cv, true>), dim3(1024), dim3(256), 0, st->s, offs, B, scfg, rawA, rawB,
                           buckets, st->heavy.as<uint32_t>());
        pt.mark();
        hipLaunchKernelGGL((k_msm_bucket_fold<Cv, true>), dim3((unsigned)((Tf + 63) / 64)), dim3(64), 0, st->s, buckets, raw,
                           offs, scfg, B, Bper, g, fold);
    } else {
        JF_HIP(hipMemsetAsync(buckets, 0, (size_t)B * XZ * 4, st->s));
        hipLaunchKernelGGL((k_msm_accumulate<Cv>), dim3((unsigned)((T + 127) / 128)), dim3(128), 0, st->s,
                           st->sorted.as<uint32_t>(), offs, B, srs->table, scfg, buckets,
                           st->partA.as<uint32_t>(), st->partB.as<uint32_t>());
        pt.mark();
        hipLaunchKernelGGL((k_msm_fixup<Cv, false>), dim3((B + 127) / 128), dim3(128), 0, st->s, offs, B, scfg,
                           st->partA.as<uint32_t>(), st->partB.as<uint32_t>(), buckets, st->heavy.as<uint32_t>());
        hipLaunchKernelGGL((k_msm_fixup_heavy_chunks<Cv, false>), dim3(1024), dim3(256), 0, st->s, offs, B, scfg,
                           st->partA.as<uint32_t>(), st->heavy.as<uint32_t>());
        hipLaunchKernelGGL((k_msm_fixup_heavy<Cv, false>), dim3(1024), dim3(256), 0, st->s, offs, B, scfg,
                           st->partA.as<uint32_t>(), st->partB.as<uint32_t>(), buckets, st->heavy.as<uint32_t>());
        pt.mark();
        // partA is free again after the fix-up
        hipLaunchKernelGGL((k_msm_bucket_fold<Cv, false>), dim3((unsigned)((Tf + 63) / 64)), dim3(64), 0, st->s, buckets,
                           buckets, offs, scfg, B, Bper, g, fold);
    }
    JF_HIP(hipGetLastError());
    pt.mark();
    if (srs->f29) JF_TRY((g1_sum_xyzz_t<Cv, true>(fold, Tf_per, d_out_xyzz, st, P)));
    else JF_TRY((g1_sum_xyzz_t<Cv, false>(fold, Tf_per, d_out_xyzz, st, P)));
    pt.mark();
    pt.finish();
    return JFHIP_OK;
}

int32_t msm_multi_device(const jfhip_srs* srs, const MsmBatch& mb, uint32_t* d_out_xyzz, jfhip_stream* st) {
    if (mb.count == 0 || mb.count > MSM_MAX_BATCH) {
        set_error("msm batch of %u polynomials (1..%d per pipeline)", mb.count, MSM_MAX_BATCH);
        return JFHIP_ERR_INVALID;
    }
    if (srs->curve == JFHIP_BN254) return msm_multi_t<Bn254G1>(srs, mb, d_out_xyzz, st);
    return msm_multi_t<Bls381G1>(srs, mb, d_out_xyzz, st);
}

int32_t msm_partial_device(const jfhip_srs* srs, size_t offset, const uint32_t* d_scalars, size_t n,
                           uint32_t* d_out_xyzz, jfhip_stream* st) {
    MsmBatch mb = {};
    mb.count = 1;
    mb.scalars[0] = d_scalars;
    mb.n[0] = (uint32_t)n;
    mb.offset[0] = (uint32_t)offset;
    if (n >= ((size_t)1 << 32)) {
        set_error("msm too large: %zu terms", n);
        return JFHIP_ERR_INVALID;
    }
    return msm_multi_device(srs, mb, d_out_xyzz, st);
}

template <class Cv>
static int32_t g1_sum_to_affine_t(const uint32_t* d_partials, size_t count, uint32_t* d_out, jfhip_stream* st) {
    constexpr int XZ = 4 * Cv::Fq::N;
    JF_TRY(st->result.ensure(XZ * 4));
    const uint32_t* src = d_partials;
    if (count != 1) {
        JF_TRY((g1_sum_xyzz_t<Cv, false>(d_partials, count, st->result.as<uint32_t>(), st)));
        src = st->result.as<uint32_t>();
    }
    hipLaunchKernelGGL((k_g1_to_affine<Cv>), dim3(1), dim3(64), 0, st->s, src, d_out, 0u, 0u);
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

// count saturated XYZZ elements (back to back) -> count records of rec_bytes (packed affine point + u32 infinity flag)
int32_t g1_batch_to_affine(int32_t curve, const uint32_t* d_xyzz, size_t count, uint32_t* d_out, size_t rec_bytes,
                           jfhip_stream* st) {
    if (count == 0) return JFHIP_OK;
    if (curve == JFHIP_BN254)
        hipLaunchKernelGGL((k_g1_to_affine<Bn254G1>), dim3((unsigned)count), dim3(64), 0, st->s, d_xyzz, d_out,
                           (uint32_t)(4 * Bn254G1::Fq::N), (uint32_t)(rec_bytes / 4));
    else
        hipLaunchKernelGGL((k_g1_to_affine<Bls381G1>), dim3((unsigned)count), dim3(64), 0, st->s, d_xyzz, d_out,
                           (uint32_t)(4 * Bls381G1::Fq::N), (uint32_t)(rec_bytes / 4));
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

int32_t g1_sum_to_affine(int32_t curve, const uint32_t* d_partials, size_t count, uint32_t* d_out,
                         jfhip_stream* st) {
    if (curve == JFHIP_BN254) return g1_sum_to_affine_t<Bn254G1>(d_partials, count, d_out, st);
    return g1_sum_to_affine_t<Bls381G1>(d_partials, count, d_out, st);
}

int32_t g1_op_device(int32_t curve, int32_t op, const uint32_t* P, const uint32_t* Q, uint32_t* out, hipStream_t s) {
    if (curve == JFHIP_BN254) hipLaunchKernelGGL((k_g1_op<Bn254G1>), dim3(1), dim3(64), 0, s, op, P, Q, out);
    else hipLaunchKernelGGL((k_g1_op<Bls381G1>), dim3(1), dim3(64), 0, s, op, P, Q, out);
    JF_HIP(hipGetLastError());
    return JFHIP_OK;
}

}  // namespace jf
