// Host-side plumbing shared by the libjfhip translation units: error reporting, streams
// with their device workspaces, phase timing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/jfhip.h"

namespace jf {

void set_error(const char* fmt, ...);
int32_t hip_fail(hipError_t e, const char* what, const char* file, int line);

#define JF_HIP(expr)                                                       \
    do {                                                                   \
        hipError_t _e = (expr);                                            \
        if (_e != hipSuccess) return jf::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define JF_TRY(expr)                 \
    do {                             \
        int32_t _rc = (expr);        \
        if (_rc != JFHIP_OK) return _rc; \
    } while (0)

// growable device buffer (grow-only; reallocation synchronises the device)
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int32_t ensure(size_t bytes);
    void release();
    template <class T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};

constexpr int MAX_PHASES = 16;

// A process-wide cache of device tables (NTT twiddle tables, quotient-domain inverses), bounded by bytes and evicted
// least-recently-used.  get() hands out the value together with a Lease: while a Lease is alive the entry cannot be
// evicted, so a caller keeps it until the kernels that read the tables are ENQUEUED (eviction frees with hipFree,
// which waits for enqueued work).  Val needs: size_t bytes() const; void release().
using Lease = std::shared_ptr<void>;
size_t cache_byte_cap();   // JFHIP_CACHE_BYTES, default 8 GiB per cache
size_t fk_cache_byte_cap();   // JFHIP_FK_CACHE_BYTES, default 4 GiB per SRS handle

template <class Key, class Val>
struct DevCache {
    // one table set being built or built.  m is held by the builder for the length of build(): threads that want the same
    // key wait on it; threads that want other keys (or other devices) never do -- the cache-wide mutex covers only the
    // map bookkeeping, not allocation, kernel launches or stream synchronisation.
    struct Slot {
        std::mutex m;
        int32_t rc = JFHIP_OK;
        Val v{};
    };
    struct Entry {
        std::shared_ptr<Slot> slot;
        uint64_t last = 0;
        int pins = 0;
        size_t bytes = 0;    // accounted size: the caller's hint while building, v.bytes() afterwards
        bool built = false;
    };
    std::mutex mu;
    std::map<Key, Entry> m;
    uint64_t tick = 0;
    size_t total = 0;

    void unpin(const Key& k) {
        std::lock_guard<std::mutex> lk(mu);
        auto it = m.find(k);
        if (it != m.end() && it->second.pins > 0) it->second.pins--;
    }
    // unlink unpinned, built entries, oldest first, until `need` more bytes fit under the cap (caller holds mu); the
    // device memory itself is freed by the caller AFTER it has dropped mu (hipFree synchronises the device)
    void make_room(size_t need, std::vector<std::shared_ptr<Slot>>& victims) {
        const size_t cap = cache_byte_cap();
        while (total + need > cap) {
            auto victim = m.end();
            for (auto it = m.begin(); it != m.end(); ++it)
                if (it->second.pins == 0 && it->second.built &&
                    (victim == m.end() || it->second.last < victim->second.last))
                    victim = it;
            if (victim == m.end()) return;  // everything is in use: grow past the cap rather than fail
            total -= victim->second.bytes;
            victims.push_back(victim->second.slot);
            m.erase(victim);
        }
    }
    template <class Build>
    int32_t get(const Key& k, Val* out, Lease* lease, size_t bytes_hint, Build build) {
        std::shared_ptr<Slot> slot;
        std::vector<std::shared_ptr<Slot>> victims;
        bool builder = false;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = m.find(k);
            if (it == m.end()) {
                make_room(bytes_hint, victims);
                Entry e;
                e.slot = std::make_shared<Slot>();
                e.bytes = bytes_hint;
                e.slot->m.lock();   // uncontended: nobody else has seen the slot yet (lock order: mu, then Slot::m)
                it = m.emplace(k, e).first;
                total += bytes_hint;
                builder = true;
            }
            it->second.last = ++tick;
            it->second.pins++;
            slot = it->second.slot;
        }
        for (auto& v : victims) v->v.release();
        if (builder) {
            slot->rc = build(&slot->v);
            if (slot->rc != JFHIP_OK) slot->v.release();  // partial allocations of a failed build
            const int32_t rc = slot->rc;
            {
                std::lock_guard<std::mutex> lk(mu);
                auto it = m.find(k);
                if (it != m.end() && it->second.slot == slot) {
                    total -= it->second.bytes;
                    if (rc == JFHIP_OK) {
                        it->second.bytes = slot->v.bytes();
                        it->second.built = true;
                        total += it->second.bytes;
                    } else {
                        m.erase(it);   // waiters hold the slot and read its rc
                    }
                }
            }
            slot->m.unlock();
            if (rc != JFHIP_OK) return rc;
        } else {
            std::lock_guard<std::mutex> wait(slot->m);   // until the builder is done
            if (slot->rc != JFHIP_OK) {
                unpin(k);
                set_error("device table build failed on another thread (status %d)", slot->rc);
                return slot->rc;
            }
        }
        *out = slot->v;
        *lease = Lease(nullptr, [this, k](void*) { this->unpin(k); });
        return JFHIP_OK;
    }
    // free every entry that no call holds
    void clear() {
        std::vector<std::shared_ptr<Slot>> victims;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (auto it = m.begin(); it != m.end();) {
                if (it->second.pins == 0 && it->second.built) {
                    total -= it->second.bytes;
                    victims.push_back(it->second.slot);
                    it = m.erase(it);
                } else ++it;
            }
        }
        for (auto& v : victims) v->v.release();
    }
};
void ntt_cache_clear();
void quotient_cache_clear();
void srs_release_fk_cache(const jfhip_srs* srs);   // ecntt.hip

}  // namespace jf

struct jfhip_stream {
    hipStream_t s = nullptr;
    bool owned = false;
    std::mutex mu;  // one MSM/NTT at a time per stream object (workspace is per stream)
    // MSM workspace
    jf::DevBuf scalars, dig, dig2, sorted, counts, offs, cursor, bsum, midbase, midcnt, midcur, finecnt, finecur, buckets, partA, partB, red0, red1, result, raw29, heavy, tilehist, tilebase;
    // NTT workspace
    jf::DevBuf ntt_tmp, ntt_io;
    // open()
    jf::DevBuf poly_a, poly_b, poly_c, quot_consts;
    // lanes for batched commits: independent MSMs run on auxiliary streams (each with its own workspace)
    // so the latency-bound reduction tail of one overlaps the bucket accumulation of the next
    static constexpr int MAX_LANES = 3;
    jfhip_stream* lane[MAX_LANES - 1] = {};
    hipEvent_t fork_ev = nullptr, join_ev[MAX_LANES - 1] = {};
    // every device buffer of the workspace (release, accounting): a new DevBuf member must be listed here only
    template <class F>
    void for_each_buf(F f) {
        jf::DevBuf* all[] = {&scalars, &dig, &dig2, &sorted, &counts, &offs, &cursor, &bsum, &midbase, &midcnt, &midcur,
                             &finecnt, &finecur, &buckets, &partA, &partB, &red0, &red1, &result, &raw29, &heavy, &tilehist, &tilebase,
                             &ntt_tmp, &ntt_io, &poly_a, &poly_b, &poly_c, &quot_consts};
        for (jf::DevBuf* b : all) f(b);
    }
    // phase timing
    hipEvent_t ev[jf::MAX_PHASES + 1] = {};
    bool ev_ready = false;
    int n_phases = 0;
    int phase_kind = 0;
};

struct jfhip_srs {
    int32_t curve = 0;
    size_t n = 0;       // points
    int32_t c = 0;      // window bits
    int32_t W = 0;      // windows
    int device = 0;
    uint32_t* table = nullptr;  // [W][n] packed affine, entry (w, i) = 2^(c*w) * P_i
    bool f29 = false;           // table coordinates carry the Montgomery factor 2^(29 L) of fp29.hpp instead of 2^(32 N)
    // window tables over a prefix of the points with window_bits chosen for THAT length, built on the first MSM that is
    // much shorter than the SRS (api.hip srs_route): the handle's own table would make it clear and fold 2^(c-1) buckets
    mutable std::mutex prefix_mu;
    mutable std::vector<jfhip_srs*> prefixes;
    // FK23 (ecntt.hip): the group-element DFT of the first d points, zero-extended to 2^log_M, depends on the SRS only;
    // kept per (d, log_M) after the first multi-open of that shape (2^log_M saturated XYZZ elements), under its own
    // mutex fk_mu (never held while a transform runs), bounded by bytes (fk_cache_byte_cap(), least recently used out)
    // and released by jfhip_srs_release_caches / jfhip_cache_clear / jfhip_srs_free
    struct FkEntry {
        uint32_t* p = nullptr;
        size_t bytes = 0;
        uint64_t last = 0;
    };
    mutable std::mutex fk_mu;
    mutable std::map<std::pair<uint32_t, uint32_t>, FkEntry> fk_dft;
    mutable size_t fk_bytes = 0;
    mutable uint64_t fk_tick = 0;
};

namespace jf {

jfhip_stream* resolve_stream(jfhip_stream* s);  // NULL -> per-thread default
bool profiling_enabled();

struct PhaseTimer {
    jfhip_stream* st;
    bool on;
    int idx = 0;
    PhaseTimer(jfhip_stream* s, int kind);
    void mark();  // records the boundary after the phase just enqueued
    void finish();
};

// The MSMs of one pipeline (msm.hip msm_multi_t): polynomial p is n[p] scalars at scalars[p] against the points from
// offset[p] on.  tile0 is filled by the pipeline (first sort tile of each polynomial).  Passed to kernels by value.
constexpr int MSM_MAX_BATCH = 16;
struct MsmBatch {
    const uint32_t* scalars[MSM_MAX_BATCH];
    uint32_t n[MSM_MAX_BATCH];
    uint32_t offset[MSM_MAX_BATCH];
    uint32_t tile0[MSM_MAX_BATCH + 1];
    uint32_t count;
};
int32_t msm_multi_device(const jfhip_srs* srs, const MsmBatch& mb, uint32_t* d_out_xyzz, jfhip_stream* st);
int32_t g1_batch_to_affine(int32_t curve, const uint32_t* d_xyzz, size_t count, uint32_t* d_out, size_t rec_bytes,
                           jfhip_stream* st);

// per-curve / per-field entry points implemented in msm.hip / ntt.hip / poly.hip
int32_t msm_partial_device(const jfhip_srs* srs, size_t offset, const uint32_t* d_scalars, size_t n,
                           uint32_t* d_out_xyzz, jfhip_stream* st);
int32_t g1_sum_to_affine(int32_t curve, const uint32_t* d_partials, size_t count, uint32_t* d_out, jfhip_stream* st);
int32_t srs_build_table(jfhip_srs* srs, const uint32_t* d_points_packed, jfhip_stream* st);
int32_t table_rescale(int32_t curve, uint32_t* d_pts, size_t count, int up, hipStream_t s);
int32_t srs_gen_points(int32_t curve, const uint64_t* beta_mont, const uint64_t* g_xy, size_t n, uint32_t* d_out,
                       jfhip_stream* st);
int32_t srs_decode_points(int32_t curve, const uint8_t* d_raw, size_t n, int compressed, int validate, uint32_t* d_out,
                          uint32_t* d_status, jfhip_stream* st);
int32_t srs_pack_points(int32_t curve, const uint8_t* d_raw, size_t n, size_t stride, uint32_t* d_out,
                        jfhip_stream* st);
int32_t ntt_device(int32_t curve, uint32_t* d_data, uint32_t log_n, size_t batch, int inverse,
                   const uint64_t* coset_mont, size_t in_len, jfhip_stream* st);
int32_t ntt_root_tables(int32_t curve, uint32_t log_n, const uint32_t** lo, const uint32_t** hi, uint32_t* a,
                        jfhip_stream* st, Lease* lease);
int32_t plonk_quotient_turbo_device(int32_t curve, uint32_t log_n, uint32_t log_m, const uint32_t* sel,
                                    const uint32_t* sig, const uint32_t* wires, const uint32_t* zperm,
                                    const uint32_t* pi, const uint64_t* k_mont, const uint64_t* challenges_mont,
                                    uint32_t* out, jfhip_stream* st);
int32_t plonk_quotient_ultra_device(int32_t curve, uint32_t log_n, uint32_t log_m, const uint32_t* sel,
                                    const uint32_t* sig, const uint32_t* wires, const uint32_t* zperm,
                                    const uint32_t* pi, const uint32_t* lookup, const uint64_t* k_mont,
                                    const uint64_t* challenges_mont, uint32_t* out, jfhip_stream* st);
int32_t plonk_quotient_coset_device(int32_t curve, int nw, uint32_t log_n, uint32_t log_m, const uint64_t* offset_mont,
                                    const uint32_t* sel, const uint32_t* sig, const uint32_t* wires,
                                    const uint32_t* zperm, const uint32_t* pi, const uint32_t* lookup,
                                    const uint64_t* k_mont, const uint64_t* challenges_mont, uint32_t* out,
                                    jfhip_stream* st, uint32_t zero_selector_mask = 0);
int32_t plonk_perm_product_device(int32_t curve, uint32_t log_n, uint32_t nw, const uint32_t* wires,
                                  const uint32_t* idp, const uint32_t* sig, const uint64_t* beta, const uint64_t* gamma,
                                  uint32_t* out, jfhip_stream* st);
int32_t plonk_lookup_product_device(int32_t curve, uint32_t log_n, const uint32_t* f, const uint32_t* t,
                                    const uint32_t* s, const uint64_t* beta, const uint64_t* gamma, uint32_t* out,
                                    jfhip_stream* st);
int32_t plonk_lookup_merge_device(int32_t curve, uint32_t log_n, const uint32_t* wires, const uint32_t* tabs,
                                  const uint64_t* tau, uint32_t* out_table, uint32_t* out_lookup, jfhip_stream* st);
int32_t open_witness_device(int32_t curve, const uint32_t* d_coeffs, size_t n, const uint32_t* d_point,
                            uint32_t* d_witness, uint32_t* d_eval, jfhip_stream* st);
int32_t noalloc_setup_device(int dev, uint32_t** out);
int32_t poly_divide_linear_noalloc(int32_t curve, const uint32_t* d_coeffs, size_t n, const uint64_t* point_mont,
                                   uint32_t* d_out, hipStream_t stream);
int32_t witness_gather_device(const uint32_t* d_witness, size_t num_vars, const uint32_t* d_wire_variables, size_t count,
                              uint32_t* d_out, jfhip_stream* st);
// n + blinders packed affine points: [L_i(beta)] g for the radix-2 domain of size n = 2^log_n, then S_(n+k) - S_k (ecntt.hip)
int32_t srs_lagrange_points(const jfhip_srs* srs, uint32_t log_n, uint32_t blinders, uint32_t* d_packed, jfhip_stream* st);
int32_t multi_open_rou_device(const jfhip_srs* srs, const uint32_t* d_coeffs, size_t ncoeffs, uint32_t log_dom,
                              size_t num_points, uint32_t* d_out, size_t rec_bytes, uint32_t log_M, jfhip_stream* st);
int32_t plonk_lookup_sorted_device(const uint32_t* d_table, size_t n, const uint32_t* d_lookups, size_t n_lookups,
                                   uint32_t* d_sorted, jfhip_stream* st);
int32_t mle_fold_device(int32_t curve, const uint32_t* d_f, size_t half, const uint32_t* d_point, uint32_t* d_q,
                        uint32_t* d_r, jfhip_stream* st);
int32_t poly_lincomb_device(int32_t curve, const uint32_t* const* d_polys, const size_t* lens, size_t count,
                            const uint64_t* coeffs_mont, uint32_t* d_out, size_t n, jfhip_stream* st);
int32_t poly_eval_batch_device(int32_t curve, const uint32_t* const* d_polys, const size_t* lens, size_t count,
                               const uint32_t* d_point, uint32_t* d_out, jfhip_stream* st);
int32_t field_op_device(int32_t field_id, int32_t op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n,
                        hipStream_t s);
int32_t g1_op_device(int32_t curve, int32_t op, const uint32_t* P, const uint32_t* Q, uint32_t* out, hipStream_t s);
int32_t probe(int32_t kind, int32_t arg, double* out);

inline int fq_limbs32(int32_t curve) { return curve == JFHIP_BN254 ? 8 : 12; }
int choose_window_bits(int32_t curve, size_t n);
int windows_for(int32_t curve, int c);

}  // namespace jf
