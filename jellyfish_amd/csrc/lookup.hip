// The Plookup sorted vector on the device: compute_lookup_sorted_vec_polynomials before its two iffts
// (reference relation/src/constraint_system.rs:1597-1620).
//
//   lookup_map[elem] = number of lookups equal to elem                      (HashMap over the n - 1 lookup values)
//   for elem in table: sorted.extend([elem; 1 + lookup_map.remove(elem).unwrap_or(0)])
//   sorted.len() != 2n - 1  ->  "The sorted vector has wrong length, some lookup variables might be outside the table"
//
// i.e. every table element in table order, followed by as many copies as it is looked up, counted at its FIRST
// occurrence in the table.  Field elements are compared as their 32 bytes (fully reduced Montgomery limbs: equal
// elements have equal bytes).  Device form, no sort:
//   1. an open-addressing table of u32 slots maps a value to the smallest table position that holds it
//      (insert = CAS on an empty slot, atomicMin on a slot whose row compares equal; a position whose left neighbour
//      holds the same value is not inserted: padded tables are one long run);
//   2. every lookup probes for its value and adds 1 to mult[first position]; lanes of a wave that found the same
//      position add once (a circuit's dummy lookups are millions of equal values: one same-address atomic costs ~80 ns);
//      a lookup that reaches an empty slot is outside the table: error flag;
//   3. exclusive scan of mult (1 + count) -> where every table position starts in the sorted vector;
//   4. every output position binary-searches its table position and copies the row.
#include "common.hpp"

namespace jf {

constexpr uint32_t LK_EMPTY = 0xffffffffu;

struct Row32 {
    uint4 a, b;
};
__device__ __forceinline__ Row32 lk_load(const uint32_t* __restrict__ base, size_t i) {
    const uint4* p = reinterpret_cast<const uint4*>(base + i * 8);
    Row32 r;
    r.a = p[0];
    r.b = p[1];
    return r;
}
__device__ __forceinline__ bool lk_eq(const Row32& x, const Row32& y) {
    return ((x.a.x ^ y.a.x) | (x.a.y ^ y.a.y) | (x.a.z ^ y.a.z) | (x.a.w ^ y.a.w) | (x.b.x ^ y.b.x) | (x.b.y ^ y.b.y) |
            (x.b.z ^ y.b.z) | (x.b.w ^ y.b.w)) == 0;
}
__device__ __forceinline__ uint32_t lk_hash(const Row32& r) {
    uint32_t h = r.a.x * 0x9E3779B1u;
    h = (h ^ (h >> 15)) + r.a.y * 0x85EBCA77u;
    h = (h ^ (h >> 13)) + r.a.z * 0xC2B2AE3Du;
    h = (h ^ (h >> 16)) + r.a.w * 0x27D4EB2Fu;
    h = (h ^ (h >> 15)) + (r.b.x ^ r.b.z) * 0x165667B1u + (r.b.y ^ r.b.w);
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    return h;
}

__global__ __launch_bounds__(256) void k_lk_insert(const uint32_t* __restrict__ table, uint32_t n, uint32_t* slots,
                                                   uint32_t mask, uint32_t* __restrict__ mult) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    mult[j] = 1u;
    const Row32 row = lk_load(table, j);
    if (j > 0 && lk_eq(lk_load(table, j - 1), row)) return;   // not a first occurrence
    uint32_t h = lk_hash(row) & mask;
    for (;;) {
        uint32_t cur = __atomic_load_n(&slots[h], __ATOMIC_RELAXED);
        if (cur == LK_EMPTY) {
            cur = atomicCAS(&slots[h], LK_EMPTY, j);
            if (cur == LK_EMPTY) return;
        }
        if (lk_eq(lk_load(table, cur), row)) {
            if (j < cur) atomicMin(&slots[h], j);
            return;
        }
        h = (h + 1u) & mask;
    }
}

__global__ __launch_bounds__(256) void k_lk_count(const uint32_t* __restrict__ table, const uint32_t* __restrict__ lookups,
                                                  uint32_t n_lookups, const uint32_t* __restrict__ slots, uint32_t mask,
                                                  uint32_t* mult, uint32_t* flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t first = LK_EMPTY;
    if (i < n_lookups) {
        const Row32 row = lk_load(lookups, i);
        uint32_t h = lk_hash(row) & mask;
        for (;;) {
            const uint32_t cur = slots[h];
            if (cur == LK_EMPTY) break;
            if (lk_eq(lk_load(table, cur), row)) {
                first = cur;
                break;
            }
            h = (h + 1u) & mask;
        }
        if (first == LK_EMPTY) atomicOr(flag, 1u);   // a lookup value outside the table
    }
    // one atomic per (wave, distinct position)
    bool active = first != LK_EMPTY;
    const uint32_t lane = threadIdx.x & 63u;
    for (;;) {
        const unsigned long long todo = __ballot(active);
        if (!todo) break;
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lv = (uint32_t)__shfl((int)first, leader, 64);
        const bool same = active && first == lv;
        const unsigned long long grp = __ballot(same);
        if (lane == (uint32_t)leader) atomicAdd(&mult[lv], (uint32_t)__popcll(grp));
        active = active && !same;
    }
}

// exclusive scan of u32 (three launches), off[m] = total
constexpr int LK_SCAN_ITEMS = 8;
__global__ __launch_bounds__(256) void k_lk_scan_local(const uint32_t* __restrict__ in, uint32_t m, uint32_t* __restrict__ out,
                                                       uint32_t* __restrict__ bsum) {
    __shared__ uint32_t sh[256];
    const uint32_t base = (blockIdx.x * 256u + threadIdx.x) * LK_SCAN_ITEMS;
    uint32_t v[LK_SCAN_ITEMS], tot = 0;
#pragma unroll
    for (int k = 0; k < LK_SCAN_ITEMS; k++) {
        v[k] = (base + k < m) ? in[base + k] : 0u;
        tot += v[k];
    }
    sh[threadIdx.x] = tot;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t t = (threadIdx.x >= (unsigned)d) ? sh[threadIdx.x - d] : 0u;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t excl = sh[threadIdx.x] - tot;
#pragma unroll
    for (int k = 0; k < LK_SCAN_ITEMS; k++) {
        if (base + k < m) out[base + k] = excl;
        excl += v[k];
    }
    if (threadIdx.x == 255) bsum[blockIdx.x] = sh[255];
}
__global__ __launch_bounds__(256) void k_lk_scan_top(uint32_t* __restrict__ bsum, uint32_t nblk, uint32_t* __restrict__ total) {
    __shared__ uint32_t sh[256];
    const uint32_t per = (nblk + 255u) / 256u, lo = threadIdx.x * per;
    uint32_t tot = 0;
    for (uint32_t k = 0; k < per; k++)
        if (lo + k < nblk) tot += bsum[lo + k];
    sh[threadIdx.x] = tot;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t t = (threadIdx.x >= (unsigned)d) ? sh[threadIdx.x - d] : 0u;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t excl = sh[threadIdx.x] - tot;
    for (uint32_t k = 0; k < per; k++)
        if (lo + k < nblk) {
            const uint32_t t = bsum[lo + k];
            bsum[lo + k] = excl;
            excl += t;
        }
    if (threadIdx.x == 255) *total = sh[255];
}
__global__ __launch_bounds__(256) void k_lk_scan_add(uint32_t* __restrict__ off, uint32_t m, const uint32_t* __restrict__ bsum) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) off[i] += bsum[i / (256u * LK_SCAN_ITEMS)];
}

// out[p] = table[j], j the table position whose span [off[j], off[j + 1]) holds p; nothing is written when the total is
// not the expected one (the caller reports the error)
__global__ __launch_bounds__(256) void k_lk_expand(const uint32_t* __restrict__ table, uint32_t n,
                                                   const uint32_t* __restrict__ off, uint32_t expected,
                                                   uint32_t* __restrict__ out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (off[n] != expected || p >= expected) return;
    uint32_t lo = 0, hi = n;   // off[lo] <= p < off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= p) lo = mid;
        else hi = mid;
    }
    const Row32 r = lk_load(table, lo);
    uint4* q = reinterpret_cast<uint4*>(out + (size_t)p * 8);
    q[0] = r.a;
    q[1] = r.b;
}

// d_table: n rows of 32 B, d_lookups: n_lookups rows, d_sorted: n + n_lookups rows.  Synchronous (the status is read back).
int32_t plonk_lookup_sorted_device(const uint32_t* d_table, size_t n, const uint32_t* d_lookups, size_t n_lookups,
                                   uint32_t* d_sorted, jfhip_stream* st) {
    if (n == 0 || n + n_lookups >= ((size_t)1 << 31)) {
        set_error("lookup sorted vector: table of %zu rows unsupported", n);
        return JFHIP_ERR_INVALID;
    }
    uint32_t cap = 2;
    while (cap < 2 * n) cap <<= 1;   // load factor <= 1/2
    const uint32_t nblk = (uint32_t)((n + 256 * LK_SCAN_ITEMS - 1) / (256 * LK_SCAN_ITEMS));
    // workspace out of the stream's MSM buffers (the stream's mutex is held by the caller)
    JF_TRY(st->dig.ensure((size_t)cap * 4));
    JF_TRY(st->finecnt.ensure(n * 4));
    JF_TRY(st->finecur.ensure((n + 1) * 4));
    JF_TRY(st->red0.ensure((size_t)nblk * 4));
    JF_TRY(st->heavy.ensure(16));
    uint32_t* slots = st->dig.as<uint32_t>();
    uint32_t* mult = st->finecnt.as<uint32_t>();
    uint32_t* off = st->finecur.as<uint32_t>();
    uint32_t* bsum = st->red0.as<uint32_t>();
    uint32_t* flag = st->heavy.as<uint32_t>();
    JF_HIP(hipMemsetAsync(slots, 0xff, (size_t)cap * 4, st->s));
    JF_HIP(hipMemsetAsync(flag, 0, 16, st->s));
    const uint32_t expected = (uint32_t)(n + n_lookups);
    hipLaunchKernelGGL(k_lk_insert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st->s, d_table, (uint32_t)n, slots, cap - 1,
                       mult);
    if (n_lookups)
        hipLaunchKernelGGL(k_lk_count, dim3((unsigned)((n_lookups + 255) / 256)), dim3(256), 0, st->s, d_table, d_lookups,
                           (uint32_t)n_lookups, slots, cap - 1, mult, flag);
    hipLaunchKernelGGL(k_lk_scan_local, dim3(nblk), dim3(256), 0, st->s, mult, (uint32_t)n, off, bsum);
    hipLaunchKernelGGL(k_lk_scan_top, dim3(1), dim3(256), 0, st->s, bsum, nblk, off + n);
    hipLaunchKernelGGL(k_lk_scan_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st->s, off, (uint32_t)n, bsum);
    hipLaunchKernelGGL(k_lk_expand, dim3((expected + 255) / 256), dim3(256), 0, st->s, d_table, (uint32_t)n, off, expected,
                       d_sorted);
    JF_HIP(hipGetLastError());
    uint32_t h_flag = 0, h_total = 0;
    JF_HIP(hipMemcpyAsync(&h_flag, flag, 4, hipMemcpyDeviceToHost, st->s));
    JF_HIP(hipMemcpyAsync(&h_total, off + n, 4, hipMemcpyDeviceToHost, st->s));
    JF_HIP(hipStreamSynchronize(st->s));
    if (h_flag || h_total != expected) {
        set_error("The sorted vector has wrong length, some lookup variables might be outside the table");
        return JFHIP_ERR_INVALID;
    }
    return JFHIP_OK;
}

}  // namespace jf
