// C ABI of libjfhip (declared in include/jfhip.h): argument checking, host<->device staging,
// stream / workspace ownership, error strings.  All arithmetic is in msm.hip / ntt.hip /
// poly.hip; nothing here computes field or group operations on the host.
#include <stdarg.h>

#include <dlfcn.h>

#include <functional>
#include <thread>

#include "common.hpp"

namespace jf {

static thread_local std::string t_err;
static bool g_profiling = false;

static void stream_release(jfhip_stream* s);

// The per-thread default stream (stream == NULL in the C ABI) and its workspace live as long as their thread: a rayon /
// std::thread worker that exits gives its device memory back.  The thread that loaded the library is left alone: its
// thread_local destructors run during process teardown, possibly after the HIP runtime has gone.
static const std::thread::id g_loader_thread = std::this_thread::get_id();
struct DefaultStreamHolder {
    jfhip_stream* st = nullptr;
    ~DefaultStreamHolder() {
        if (st && std::this_thread::get_id() != g_loader_thread) {
            (void)hipStreamSynchronize(st->s);
            stream_release(st);
            delete st;
        }
        st = nullptr;
    }
};
static thread_local DefaultStreamHolder t_default;

size_t cache_byte_cap() {
    static const size_t cap = [] {
        const char* e = getenv("JFHIP_CACHE_BYTES");
        unsigned long long v = e ? strtoull(e, nullptr, 10) : 0ull;
        return v ? (size_t)v : (size_t)8 << 30;
    }();
    return cap;
}

// ---- device markers (roctx ranges) ---------------------------------------------------------------------------------
// With JFHIP_MARKERS=1 the library brackets its entry points with roctx ranges named after the reference's own timers
// (pcs/src/univariate_kzg/mod.rs:96-123 "Committing to polynomial of degree {}", :134 "batch commit {} polynomials",
// :158 "Opening polynomial of degree {}", :426-430), so a `rocprofv3 --kernel-trace --marker-trace` run of a proof is
// segmented by name instead of by kernel-name heuristics.  The roctx library is looked up at run time (rocprofv3's own,
// then roctracer's); without it, or with the variable unset, the ranges are no-ops and nothing is loaded.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
static const Roctx& roctx() {
    static const Roctx r = [] {
        Roctx x;
        const char* e = getenv("JFHIP_MARKERS");
        if (!e || e[0] == '0') return x;
        for (const char* lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4",
                                "libroctx64.so"}) {
            void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            x.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
            x.pop = (int (*)())dlsym(h, "roctxRangePop");
            if (x.push && x.pop) return x;
            x.push = nullptr;
            x.pop = nullptr;
        }
        return x;
    }();
    return r;
}
struct MarkerRange {
    bool on = false;
    explicit MarkerRange(const char* fmt, ...) {
        const Roctx& r = roctx();
        if (!r.push) return;
        char buf[160];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        r.push(buf);
        on = true;
    }
    ~MarkerRange() {
        if (on) roctx().pop();
    }
};

size_t fk_cache_byte_cap() {
    static const size_t cap = [] {
        const char* e = getenv("JFHIP_FK_CACHE_BYTES");
        unsigned long long v = e ? strtoull(e, nullptr, 10) : 0ull;
        return e ? (size_t)v : (size_t)4 << 30;
    }();
    return cap;
}

// every live SRS handle, so that jfhip_cache_clear can reach the caches the handles own
static std::mutex g_srs_mu;
static std::vector<jfhip_srs*> g_srs_live;
static void srs_register(jfhip_srs* s) {
    std::lock_guard<std::mutex> lk(g_srs_mu);
    g_srs_live.push_back(s);
}
static void srs_unregister(jfhip_srs* s) {
    std::lock_guard<std::mutex> lk(g_srs_mu);
    for (size_t i = 0; i < g_srs_live.size(); i++)
        if (g_srs_live[i] == s) {
            g_srs_live[i] = g_srs_live.back();
            g_srs_live.pop_back();
            break;
        }
}

void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    t_err = buf;
}

int32_t hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    if (e == hipErrorOutOfMemory) return JFHIP_ERR_OOM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return JFHIP_ERR_NO_DEVICE;
    return JFHIP_ERR_DEVICE;
}

int32_t DevBuf::ensure(size_t bytes) {
    if (bytes <= cap && p) return JFHIP_OK;
    if (bytes == 0) bytes = 16;
    if (p) {
        JF_HIP(hipDeviceSynchronize());
        JF_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
    }
    // round up so slightly larger follow-up problems reuse the allocation
    size_t want = (bytes + (bytes >> 3) + 255) & ~(size_t)255;
    JF_HIP(hipMalloc(&p, want));
    cap = want;
    return JFHIP_OK;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

bool profiling_enabled() { return g_profiling; }

PhaseTimer::PhaseTimer(jfhip_stream* s, int kind) : st(s), on(g_profiling) {
    if (!on) return;
    if (!st->ev_ready) {
        for (int i = 0; i <= MAX_PHASES; i++)
            if (hipEventCreate(&st->ev[i]) != hipSuccess) {
                on = false;
                return;
            }
        st->ev_ready = true;
    }
    st->phase_kind = kind;
    st->n_phases = 0;
    (void)hipEventRecord(st->ev[0], st->s);
}
void PhaseTimer::mark() {
    if (!on || idx >= MAX_PHASES) return;
    idx++;
    (void)hipEventRecord(st->ev[idx], st->s);
}
void PhaseTimer::finish() {
    if (!on) return;
    st->n_phases = idx;
}

jfhip_stream* resolve_stream(jfhip_stream* s) {
    if (s) return s;
    if (!t_default.st) {
        jfhip_stream* st = new jfhip_stream();
        if (hipStreamCreateWithFlags(&st->s, hipStreamNonBlocking) != hipSuccess) {
            delete st;
            return nullptr;
        }
        st->owned = true;
        t_default.st = st;
    }
    return t_default.st;
}

static void lanes_release(jfhip_stream* s) {
    for (int i = 0; i < jfhip_stream::MAX_LANES - 1; i++) {
        if (s->lane[i]) {
            (void)hipStreamSynchronize(s->lane[i]->s);
            stream_release(s->lane[i]);
            delete s->lane[i];
            s->lane[i] = nullptr;
        }
        if (s->join_ev[i]) (void)hipEventDestroy(s->join_ev[i]);
        s->join_ev[i] = nullptr;
    }
    if (s->fork_ev) (void)hipEventDestroy(s->fork_ev);
    s->fork_ev = nullptr;
}

static void workspace_release(jfhip_stream* s) {
    for (int i = 0; i < jfhip_stream::MAX_LANES - 1; i++)
        if (s->lane[i]) {
            (void)hipStreamSynchronize(s->lane[i]->s);
            workspace_release(s->lane[i]);
        }
    s->for_each_buf([](DevBuf* b) { b->release(); });
}

static void stream_release(jfhip_stream* s) {
    lanes_release(s);
    s->for_each_buf([](DevBuf* b) { b->release(); });
    if (s->ev_ready)
        for (int i = 0; i <= MAX_PHASES; i++) (void)hipEventDestroy(s->ev[i]);
    s->ev_ready = false;
    if (s->owned && s->s) (void)hipStreamDestroy(s->s);
    s->s = nullptr;
}

static int fq_bytes(int32_t curve) { return fq_limbs32(curve) * 4; }

static int32_t check_curve(int32_t curve) {
    if (curve != JFHIP_BN254 && curve != JFHIP_BLS12_381) {
        set_error("unknown curve id %d", curve);
        return JFHIP_ERR_INVALID;
    }
    return JFHIP_OK;
}

static int32_t srs_alloc(int32_t curve, size_t n, int32_t window_bits, jfhip_srs** out) {
    JF_TRY(check_curve(curve));
    if (n == 0) {
        set_error("SRS must hold at least one point");
        return JFHIP_ERR_INVALID;
    }
    int c = window_bits > 0 ? window_bits : choose_window_bits(curve, n);
    if (c < 2 || c > 24) {
        set_error("window_bits %d out of range [2, 24]", c);
        return JFHIP_ERR_INVALID;
    }
    int W = windows_for(curve, c);
    if ((double)W * (double)n >= 2147483648.0) {
        set_error("SRS too large: %zu points x %d windows exceeds 2^31 table entries", n, W);
        return JFHIP_ERR_INVALID;
    }
    jfhip_srs* s = new jfhip_srs();
    s->curve = curve;
    s->n = n;
    s->c = c;
    s->W = W;
    if (hipGetDevice(&s->device) != hipSuccess) s->device = 0;
    hipError_t e = hipMalloc(&s->table, (size_t)W * n * 2 * fq_bytes(curve));
    if (e != hipSuccess) {
        delete s;
        return hip_fail(e, "hipMalloc(SRS table)", __FILE__, __LINE__);
    }
    srs_register(s);
    *out = s;
    return JFHIP_OK;
}

}  // namespace jf

using namespace jf;

#define JF_STREAM(st, s)                                  \
    jfhip_stream* st = jf::resolve_stream(s);             \
    if (!st) {                                            \
        jf::set_error("could not create a HIP stream");   \
        return JFHIP_ERR_DEVICE;                          \
    }

extern "C" {

int32_t jfhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int32_t jfhip_init(int32_t device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        set_error("no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return JFHIP_ERR_NO_DEVICE;
    }
    if (device >= 0) {
        if (device >= n) {
            set_error("device %d out of range (%d visible)", device, n);
            return JFHIP_ERR_NO_DEVICE;
        }
        JF_HIP(hipSetDevice(device));
    }
    JF_HIP(hipFree(nullptr));
    // set-up of the no-allocation entry point: this device's address of its static workspace
    int cur = 0;
    JF_HIP(hipGetDevice(&cur));
    JF_TRY(noalloc_setup_device(cur, nullptr));
    return JFHIP_OK;
}

int32_t jfhip_last_error(char* buf, size_t len) {
    if (!buf || len == 0) return JFHIP_ERR_INVALID;
    snprintf(buf, len, "%s", t_err.c_str());
    return JFHIP_OK;
}

int32_t jfhip_stream_create(jfhip_stream** out) {
    if (!out) return JFHIP_ERR_INVALID;
    jfhip_stream* st = new jfhip_stream();
    hipError_t e = hipStreamCreateWithFlags(&st->s, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete st;
        return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__);
    }
    st->owned = true;
    // warm the allocator as warmup_new_stream() does (pcs/src/lib.rs:430-435)
    if (st->result.ensure(4096) != JFHIP_OK) {
        stream_release(st);
        delete st;
        return JFHIP_ERR_DEVICE;
    }
    *out = st;
    return JFHIP_OK;
}

int32_t jfhip_stream_wrap(void* hip_stream, jfhip_stream** out) {
    if (!out) return JFHIP_ERR_INVALID;
    jfhip_stream* st = new jfhip_stream();
    st->s = (hipStream_t)hip_stream;
    st->owned = false;
    *out = st;
    return JFHIP_OK;
}

int32_t jfhip_stream_destroy(jfhip_stream* s) {
    if (!s) return JFHIP_OK;
    (void)hipStreamSynchronize(s->s);
    stream_release(s);
    delete s;
    return JFHIP_OK;
}

int32_t jfhip_stream_release_workspace(jfhip_stream* s) {
    JF_STREAM(st, s);
    std::lock_guard<std::mutex> lk(st->mu);
    JF_HIP(hipStreamSynchronize(st->s));
    workspace_release(st);
    return JFHIP_OK;
}

int32_t jfhip_cache_clear(void) {
    ntt_cache_clear();
    quotient_cache_clear();
    // the FK23 transforms the live SRS handles keep (up to fk_cache_byte_cap() each)
    std::lock_guard<std::mutex> lk(g_srs_mu);
    for (jfhip_srs* s : g_srs_live) {
        srs_release_fk_cache(s);
        std::lock_guard<std::mutex> pl(s->prefix_mu);
        for (jfhip_srs* p : s->prefixes) srs_release_fk_cache(p);
    }
    return JFHIP_OK;
}

int32_t jfhip_srs_release_caches(jfhip_srs* srs) {
    if (!srs) return JFHIP_OK;
    srs_release_fk_cache(srs);
    std::lock_guard<std::mutex> pl(srs->prefix_mu);
    for (jfhip_srs* p : srs->prefixes) srs_release_fk_cache(p);
    return JFHIP_OK;
}

int32_t jfhip_shutdown(void) {
    // the calling thread's default stream and the process-wide table caches; explicit handles stay the caller's
    if (t_default.st) {
        (void)hipStreamSynchronize(t_default.st->s);
        stream_release(t_default.st);
        delete t_default.st;
        t_default.st = nullptr;
    }
    return jfhip_cache_clear();
}

int32_t jfhip_stream_sync(jfhip_stream* s) {
    JF_STREAM(st, s);
    JF_HIP(hipStreamSynchronize(st->s));
    return JFHIP_OK;
}

void* jfhip_stream_raw(jfhip_stream* s) {
    jfhip_stream* st = resolve_stream(s);
    return st ? (void*)st->s : nullptr;
}

// ---- SRS ----------------------------------------------------------------------------------------

int32_t jfhip_srs_load(int32_t curve_id, const void* points, size_t n, size_t stride_bytes, int32_t window_bits,
                       jfhip_srs** out) {
    if (!points || !out) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_TRY(check_curve(curve_id));
    const size_t pt = 2 * (size_t)fq_bytes(curve_id);
    if (stride_bytes < pt) {
        set_error("stride %zu smaller than a packed point (%zu bytes)", stride_bytes, pt);
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, nullptr);
    std::lock_guard<std::mutex> lk(st->mu);
    jfhip_srs* srs = nullptr;
    JF_TRY(srs_alloc(curve_id, n, window_bits, &srs));
    DevBuf raw, packed;
    int32_t rc = raw.ensure(n * stride_bytes);
    if (rc == JFHIP_OK) rc = packed.ensure(n * pt);
    if (rc == JFHIP_OK && hipMemcpyAsync(raw.p, points, n * stride_bytes, hipMemcpyHostToDevice, st->s) != hipSuccess)
        rc = JFHIP_ERR_DEVICE;
    if (rc == JFHIP_OK) rc = srs_pack_points(curve_id, raw.as<uint8_t>(), n, stride_bytes, packed.as<uint32_t>(), st);
    if (rc == JFHIP_OK) rc = srs_build_table(srs, packed.as<uint32_t>(), st);
    if (rc == JFHIP_OK && hipStreamSynchronize(st->s) != hipSuccess) rc = JFHIP_ERR_DEVICE;
    raw.release();
    packed.release();
    if (rc != JFHIP_OK) {
        jfhip_srs_free(srs);
        return rc;
    }
    *out = srs;
    return JFHIP_OK;
}

int32_t jfhip_srs_load_device(int32_t curve_id, const void* d_points_packed, size_t n, int32_t window_bits,
                              jfhip_srs** out) {
    if (!d_points_packed || !out) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, nullptr);
    std::lock_guard<std::mutex> lk(st->mu);
    jfhip_srs* srs = nullptr;
    JF_TRY(srs_alloc(curve_id, n, window_bits, &srs));
    int32_t rc = srs_build_table(srs, (const uint32_t*)d_points_packed, st);
    if (rc == JFHIP_OK && hipStreamSynchronize(st->s) != hipSuccess) rc = JFHIP_ERR_DEVICE;
    if (rc != JFHIP_OK) {
        jfhip_srs_free(srs);
        return rc;
    }
    *out = srs;
    return JFHIP_OK;
}

int32_t jfhip_srs_lagrange(const jfhip_srs* srs, uint32_t log_n, uint32_t blinders, int32_t window_bits, jfhip_srs** out) {
    if (!srs || !out) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    const uint32_t adicity = srs->curve == JFHIP_BN254 ? 28u : 32u;
    if (log_n > adicity || log_n > 26 || blinders > 16) {
        set_error("lagrange key: domain of 2^%u points / %u blinders unsupported", log_n, blinders);
        return JFHIP_ERR_INVALID;
    }
    const size_t n = (size_t)1 << log_n;
    if (srs->n < n + blinders) {
        set_error("lagrange key: the SRS holds %zu points, a domain of %zu with %u blinders needs %zu", srs->n, n, blinders,
                  n + blinders);
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, nullptr);
    std::lock_guard<std::mutex> lk(st->mu);
    DevBuf packed;
    JF_TRY(packed.ensure((n + blinders) * 2 * (size_t)fq_bytes(srs->curve)));
    int32_t rc = srs_lagrange_points(srs, log_n, blinders, packed.as<uint32_t>(), st);
    jfhip_srs* lag = nullptr;
    if (rc == JFHIP_OK) rc = srs_alloc(srs->curve, n + blinders, window_bits, &lag);
    if (rc == JFHIP_OK) rc = srs_build_table(lag, packed.as<uint32_t>(), st);
    if (rc == JFHIP_OK && hipStreamSynchronize(st->s) != hipSuccess) rc = JFHIP_ERR_DEVICE;
    packed.release();
    if (rc != JFHIP_OK) {
        if (lag) jfhip_srs_free(lag);
        return rc;
    }
    *out = lag;
    return JFHIP_OK;
}

int32_t jfhip_srs_load_serialized(int32_t curve_id, const uint8_t* bytes, size_t len, size_t n, int32_t compressed,
                                  int32_t validate, int32_t window_bits, jfhip_srs** out) {
    if (!bytes || !out) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_TRY(check_curve(curve_id));
    const size_t fb = (size_t)fq_bytes(curve_id), rec = compressed ? fb : 2 * fb, pt = 2 * fb;
    if (n == 0 || len / rec < n) {
        set_error("%zu bytes hold %zu %s G1 records of %zu bytes, %zu requested", len, len / rec,
                  compressed ? "compressed" : "uncompressed", rec, n);
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, nullptr);
    std::lock_guard<std::mutex> lk(st->mu);
    jfhip_srs* srs = nullptr;
    JF_TRY(srs_alloc(curve_id, n, window_bits, &srs));
    DevBuf raw, packed, status;
    uint32_t h_status[2] = {0u, 0xffffffffu};
    int32_t rc = raw.ensure(n * rec);
    if (rc == JFHIP_OK) rc = packed.ensure(n * pt);
    if (rc == JFHIP_OK) rc = status.ensure(8);
    if (rc == JFHIP_OK && (hipMemcpyAsync(raw.p, bytes, n * rec, hipMemcpyHostToDevice, st->s) != hipSuccess ||
                           hipMemcpyAsync(status.p, h_status, 8, hipMemcpyHostToDevice, st->s) != hipSuccess))
        rc = JFHIP_ERR_DEVICE;
    if (rc == JFHIP_OK)
        rc = srs_decode_points(curve_id, raw.as<uint8_t>(), n, compressed != 0, validate != 0, packed.as<uint32_t>(),
                               status.as<uint32_t>(), st);
    if (rc == JFHIP_OK && (hipMemcpyAsync(h_status, status.p, 8, hipMemcpyDeviceToHost, st->s) != hipSuccess ||
                           hipStreamSynchronize(st->s) != hipSuccess))
        rc = JFHIP_ERR_DEVICE;
    if (rc == JFHIP_OK && h_status[0] != 0) {
        // SerializationError::InvalidData of ark-serialize; the shim maps it to PCSError::InvalidParameters
        set_error("G1 record %u is invalid:%s%s%s%s", h_status[1], (h_status[0] & 8) ? " malformed flags" : "",
                  (h_status[0] & 1) ? " coordinate not below the modulus" : "",
                  (h_status[0] & 2) ? " not on the curve" : "", (h_status[0] & 4) ? " not in the prime-order subgroup" : "");
        rc = JFHIP_ERR_INVALID;
    }
    if (rc == JFHIP_OK) rc = srs_build_table(srs, packed.as<uint32_t>(), st);
    if (rc == JFHIP_OK && hipStreamSynchronize(st->s) != hipSuccess) rc = JFHIP_ERR_DEVICE;
    raw.release();
    packed.release();
    status.release();
    if (rc != JFHIP_OK) {
        jfhip_srs_free(srs);
        return rc;
    }
    *out = srs;
    return JFHIP_OK;
}

int32_t jfhip_srs_gen_for_testing(int32_t curve_id, const uint64_t* beta_mont, const uint64_t* g_xy, size_t n,
                                  int32_t window_bits, jfhip_srs** out) {
    if (!beta_mont || !g_xy || !out) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, nullptr);
    std::lock_guard<std::mutex> lk(st->mu);
    jfhip_srs* srs = nullptr;
    JF_TRY(srs_alloc(curve_id, n, window_bits, &srs));
    // generate straight into window 0 of the table, then extend
    int32_t rc = srs_gen_points(curve_id, beta_mont, g_xy, n, srs->table, st);
    if (rc == JFHIP_OK) rc = srs_build_table(srs, srs->table, st);
    if (rc == JFHIP_OK && hipStreamSynchronize(st->s) != hipSuccess) rc = JFHIP_ERR_DEVICE;
    if (rc != JFHIP_OK) {
        jfhip_srs_free(srs);
        return rc;
    }
    *out = srs;
    return JFHIP_OK;
}

int32_t jfhip_srs_free(jfhip_srs* srs) {
    if (!srs) return JFHIP_OK;
    srs_unregister(srs);
    for (jfhip_srs* p : srs->prefixes) (void)jfhip_srs_free(p);
    srs_release_fk_cache(srs);
    if (srs->table) (void)hipFree(srs->table);
    delete srs;
    return JFHIP_OK;
}

size_t jfhip_srs_len(const jfhip_srs* srs) { return srs ? srs->n : 0; }
int32_t jfhip_srs_window_bits(const jfhip_srs* srs) { return srs ? srs->c : 0; }

int32_t jfhip_srs_read(const jfhip_srs* srs, size_t offset, size_t n, void* out) {
    if (!srs || !out || offset + n > srs->n) {
        set_error("srs_read: bad range");
        return JFHIP_ERR_INVALID;
    }
    const size_t pt = 2 * (size_t)fq_bytes(srs->curve);
    if (!srs->f29 || n == 0) {
        JF_HIP(hipMemcpy(out, (const uint8_t*)srs->table + offset * pt, n * pt, hipMemcpyDeviceToHost));
        return JFHIP_OK;
    }
    // the table holds the kernels' internal Montgomery factor: convert a copy back to the boundary's form
    void* tmp = nullptr;
    JF_HIP(hipMalloc(&tmp, n * pt));
    hipError_t e = hipMemcpy(tmp, (const uint8_t*)srs->table + offset * pt, n * pt, hipMemcpyDeviceToDevice);
    int32_t rc = JFHIP_OK;
    if (e == hipSuccess) rc = table_rescale(srs->curve, (uint32_t*)tmp, n, 0, 0);
    if (e == hipSuccess && rc == JFHIP_OK) e = hipMemcpy(out, tmp, n * pt, hipMemcpyDeviceToHost);
    (void)hipFree(tmp);
    if (rc != JFHIP_OK) return rc;
    JF_HIP(e);
    return JFHIP_OK;
}

// ---- MSM ----------------------------------------------------------------------------------------

static int32_t check_msm_range(const jfhip_srs* srs, size_t offset, size_t n) {
    if (!srs) {
        set_error("null SRS handle");
        return JFHIP_ERR_INVALID;
    }
    if (offset > srs->n || n > srs->n - offset) {
        // UnivariateKzgPCS::commit: "poly degree {} is larger than allowed {}" (mod.rs:100-106)
        set_error("poly degree %zu is larger than allowed %zu", offset + n ? offset + n - 1 : 0, srs->n);
        return JFHIP_ERR_INVALID;
    }
    return JFHIP_OK;
}

// An MSM over points [offset, offset + n) with offset + n <= srs->n / 16 runs on a window table of its own over the
// first 4096 * 4^k >= offset + n points, built on the first such call and kept with the handle (all levels together
// stay below ~1/8 of the main table).  The handle's window_bits are chosen for srs->n: a 2^12-term commitment against a
// 2^24-point universal SRS would clear, scan and fold 2^21 buckets for 4096 x 12 additions.  JFHIP_SRS_PREFIX=0 turns
// it off.  If the table cannot be built (out of memory) the MSM runs on the main table as before.
static const jfhip_srs* srs_route(const jfhip_srs* srs, size_t end, jfhip_stream* st) {
    static const bool on = [] {
        const char* e = getenv("JFHIP_SRS_PREFIX");
        return !(e && e[0] == '0');
    }();
    if (!on || srs->n < ((size_t)1 << 14) || end == 0 || end > srs->n / 16) return srs;
    size_t want = 4096;
    while (want < end) want <<= 2;
    if (want > srs->n / 4) return srs;
    std::lock_guard<std::mutex> lk(srs->prefix_mu);
    for (const jfhip_srs* p : srs->prefixes)
        if (p->n == want) return p;
    jfhip_srs* p = nullptr;
    if (srs_alloc(srs->curve, want, 0, &p) != JFHIP_OK) {
        (void)hipGetLastError();  // the failed allocation must not surface at the MSM's next launch check
        return srs;
    }
    p->device = srs->device;
    const size_t pt = 2 * (size_t)fq_bytes(srs->curve);
    // window 0 of the main table = the points themselves, possibly with the kernels' internal Montgomery factor
    bool ok = hipMemcpyAsync(p->table, srs->table, want * pt, hipMemcpyDeviceToDevice, st->s) == hipSuccess;
    if (ok && srs->f29) ok = table_rescale(srs->curve, p->table, want, 0, st->s) == JFHIP_OK;
    if (ok) ok = srs_build_table(p, p->table, st) == JFHIP_OK;
    // other host threads pick the table up without any ordering on this stream
    if (ok) ok = hipStreamSynchronize(st->s) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        (void)jfhip_srs_free(p);
        return srs;
    }
    srs->prefixes.push_back(p);
    return p;
}

int32_t jfhip_msm_partial_device(const jfhip_srs* srs, size_t offset, const void* d_scalars, size_t n,
                                 void* d_out_xyzz, jfhip_stream* stream) {
    JF_TRY(check_msm_range(srs, offset, n));
    if ((n && !d_scalars) || !d_out_xyzz) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    return msm_partial_device(srs_route(srs, offset + n, st), offset, (const uint32_t*)d_scalars, n,
                              (uint32_t*)d_out_xyzz, st);
}

static int32_t msm_batch_locked(const jfhip_srs* srs, const size_t* lens, size_t batch, uint8_t* d_res, size_t rec,
                                jfhip_stream* st,
                                const std::function<int32_t(size_t, jfhip_stream*, const void**)>& stage,
                                const std::function<const jfhip_srs*(size_t)>& srs_of = nullptr,
                                const size_t* offsets = nullptr);

static int32_t msm_device_locked(const jfhip_srs* srs, size_t offset, const void* d_scalars, size_t n, void* d_out,
                                 jfhip_stream* st) {
    const size_t xz = 4 * (size_t)fq_bytes(srs->curve);
    // Experiment switch (JFHIP_MSM_DEVICE_CHUNKS=2..4, default off): a large MSM as partial sums over point ranges on the
    // lane streams, so that the bucket sort of one range (memory-bound) overlaps the bucket accumulation of the previous
    // one (VALU-bound); the price is one more fix-up / fold / sum tree per range.
    static const int dev_chunks = getenv("JFHIP_MSM_DEVICE_CHUNKS") ? atoi(getenv("JFHIP_MSM_DEVICE_CHUNKS")) : 1;
    if (dev_chunks >= 2 && dev_chunks <= 4 && n >= ((size_t)1 << 22)) {
        const size_t chunks = (size_t)dev_chunks;
        JF_TRY(st->result.ensure(chunks * xz + 256));
        std::vector<size_t> lens(chunks), offs(chunks);
        const size_t per = (n + chunks - 1) / chunks;
        for (size_t b = 0; b < chunks; b++) {
            const size_t lo = b * per < n ? b * per : n, hi = (b + 1) * per < n ? (b + 1) * per : n;
            offs[b] = offset + lo;
            lens[b] = hi - lo;
        }
        const uint8_t* base = (const uint8_t*)d_scalars;
        JF_TRY(msm_batch_locked(
            srs, lens.data(), chunks, st->result.as<uint8_t>(), xz, st,
            [&](size_t b, jfhip_stream*, const void** out) -> int32_t {
                *out = base + (offs[b] - offset) * 32;
                return JFHIP_OK;
            },
            nullptr, offs.data()));
        return g1_sum_to_affine(srs->curve, st->result.as<uint32_t>(), chunks, (uint32_t*)d_out, st);
    }
    JF_TRY(st->result.ensure(xz + 256));
    JF_TRY(msm_partial_device(srs_route(srs, offset + n, st), offset, (const uint32_t*)d_scalars, n,
                              st->result.as<uint32_t>(), st));
    return g1_sum_to_affine(srs->curve, st->result.as<uint32_t>(), 1, (uint32_t*)d_out, st);
}

int32_t jfhip_msm_device(const jfhip_srs* srs, size_t offset, const void* d_scalars, size_t n, void* d_out,
                         jfhip_stream* stream) {
    MarkerRange mr("Committing to polynomial of degree %zu (device scalars)", n ? n - 1 : 0);
    JF_TRY(check_msm_range(srs, offset, n));
    if ((n && !d_scalars) || !d_out) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    return msm_device_locked(srs, offset, d_scalars, n, d_out, st);
}

int32_t jfhip_g1_sum_partials(int32_t curve_id, const void* d_partials, size_t count, void* d_out,
                              jfhip_stream* stream) {
    JF_TRY(check_curve(curve_id));
    if ((count && !d_partials) || !d_out) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    if (count == 0) {
        const size_t bytes = 2 * (size_t)fq_bytes(curve_id);
        JF_HIP(hipMemsetAsync(d_out, 0, bytes, st->s));
        uint32_t one = 1;
        JF_HIP(hipMemcpyAsync((uint8_t*)d_out + bytes, &one, 4, hipMemcpyHostToDevice, st->s));
        JF_HIP(hipStreamSynchronize(st->s));
        return JFHIP_OK;
    }
    return g1_sum_to_affine(curve_id, (const uint32_t*)d_partials, count, (uint32_t*)d_out, st);
}

// Independent MSMs of a batch, fanned out over up to MAX_LANES streams that fork from / join into st->s.
// `stage` (optional) uploads the scalars of entry b on the lane's stream and returns their device pointer.
// `srs_of` (optional): entry b commits against its own SRS handle (the levels of a multilinear parameter) instead of srs.
// `offsets` (optional): entry b is the partial sum over points [offsets[b], offsets[b] + lens[b]), written as one
// saturated XYZZ element (rec bytes apart) instead of a normalised point: the chunks of one MSM whose scalars arrive
// from the host (jfhip_msm).
static int32_t msm_batch_locked(const jfhip_srs* srs, const size_t* lens, size_t batch, uint8_t* d_res, size_t rec,
                                jfhip_stream* st,
                                const std::function<int32_t(size_t, jfhip_stream*, const void**)>& stage,
                                const std::function<const jfhip_srs*(size_t)>& srs_of, const size_t* offsets) {
    size_t maxwork = 0;
    for (size_t b = 0; b < batch; b++) {
        const size_t w = lens[b] * (size_t)(srs_of ? srs_of(b) : srs)->W;
        maxwork = w > maxwork ? w : maxwork;
    }
    // lanes only pay off while the tails are a visible fraction of an MSM; each lane holds a workspace
    int lanes = 1;
    static const bool force_lanes = getenv("JFHIP_MSM_LANES_FORCE") && getenv("JFHIP_MSM_LANES_FORCE")[0] == '1';
    if (batch > 1 && (maxwork <= ((size_t)1 << 26) || force_lanes)) lanes = batch < 3 ? (int)batch : 3;
    if (lanes > 1) {
        if (!st->fork_ev) JF_HIP(hipEventCreateWithFlags(&st->fork_ev, hipEventDisableTiming));
        for (int i = 0; i < lanes - 1; i++) {
            if (!st->lane[i]) {
                jfhip_stream* ls = new jfhip_stream();
                hipError_t e = hipStreamCreateWithFlags(&ls->s, hipStreamNonBlocking);
                if (e != hipSuccess) {
                    delete ls;
                    return hip_fail(e, "hipStreamCreate(lane)", __FILE__, __LINE__);
                }
                ls->owned = true;
                st->lane[i] = ls;
            }
            if (!st->join_ev[i]) JF_HIP(hipEventCreateWithFlags(&st->join_ev[i], hipEventDisableTiming));
        }
        JF_HIP(hipEventRecord(st->fork_ev, st->s));
        for (int i = 0; i < lanes - 1; i++) JF_HIP(hipStreamWaitEvent(st->lane[i]->s, st->fork_ev, 0));
    }
    int32_t rc = JFHIP_OK;
    for (size_t b = 0; b < batch && rc == JFHIP_OK; b++) {
        jfhip_stream* ls = (b % lanes == 0) ? st : st->lane[b % lanes - 1];
        const void* d_scalars = nullptr;
        rc = stage(b, ls, &d_scalars);
        if (rc != JFHIP_OK) break;
        const jfhip_srs* sb = srs_of ? srs_of(b) : srs;
        if (offsets)
            rc = msm_partial_device(srs_route(sb, offsets[b] + lens[b], ls), offsets[b], (const uint32_t*)d_scalars, lens[b],
                                    (uint32_t*)(d_res + b * rec), ls);
        else
            rc = msm_device_locked(sb, 0, d_scalars, lens[b], d_res + b * rec, ls);
    }
    // join the lanes whatever happened: the caller may release its buffers as soon as this returns an error, and
    // the lanes may still be reading / writing them
    for (int i = 0; i < lanes - 1; i++) {
        if (rc != JFHIP_OK) {
            (void)hipStreamSynchronize(st->lane[i]->s);
            continue;
        }
        JF_HIP(hipEventRecord(st->join_ev[i], st->lane[i]->s));
        JF_HIP(hipStreamWaitEvent(st->s, st->join_ev[i], 0));
    }
    if (rc != JFHIP_OK) (void)hipStreamSynchronize(st->s);
    return rc;
}

// A batch against ONE SRS as fused pipelines of up to MSM_MAX_BATCH polynomials (msm.hip msm_multi_t): one sort, one
// accumulation and one tail per pipeline instead of one per polynomial.  offsets == NULL: every polynomial starts at
// point 0.  d_xyzz != NULL: the un-normalised partial sums (saturated XYZZ, back to back) instead of records in d_res.
// Returns false (nothing enqueued) when the batch does not qualify and the caller should take the lane path.
static bool msm_batch_fusable(const jfhip_srs* srs, const size_t* lens, size_t batch) {
    static const bool on = [] {
        const char* e = getenv("JFHIP_MSM_FUSED");
        return !(e && e[0] == '0');
    }();
    if (!on || batch < 2) return false;
    // the widened sort of a pipeline holds every polynomial's entries at once (20 B each): cap it at 2^30 entries
    for (size_t b0 = 0; b0 < batch; b0 += MSM_MAX_BATCH) {
        size_t terms = 0;
        for (size_t b = b0; b < batch && b < b0 + MSM_MAX_BATCH; b++) terms += lens[b];
        if (terms * (size_t)srs->W > ((size_t)1 << 30)) return false;
    }
    return true;
}
static int32_t msm_batch_fused_locked(const jfhip_srs* srs, const size_t* offsets, const void* const* d_scalars,
                                      const size_t* lens, size_t batch, uint8_t* d_res, size_t rec, uint8_t* d_xyzz,
                                      jfhip_stream* st) {
    const size_t xz = 4 * (size_t)fq_bytes(srs->curve);
    size_t end = 0;
    for (size_t b = 0; b < batch; b++) {
        const size_t e = (offsets ? offsets[b] : 0) + lens[b];
        end = e > end ? e : end;
    }
    const jfhip_srs* table = srs_route(srs, end, st);
    if (!d_xyzz) {
        JF_TRY(st->result.ensure(batch * xz + 256));
        d_xyzz = st->result.as<uint8_t>();
    }
    for (size_t b0 = 0; b0 < batch; b0 += MSM_MAX_BATCH) {
        MsmBatch mb = {};
        mb.count = (uint32_t)(batch - b0 < (size_t)MSM_MAX_BATCH ? batch - b0 : (size_t)MSM_MAX_BATCH);
        for (uint32_t k = 0; k < mb.count; k++) {
            mb.scalars[k] = (const uint32_t*)d_scalars[b0 + k];
            mb.n[k] = (uint32_t)lens[b0 + k];
            mb.offset[k] = (uint32_t)(offsets ? offsets[b0 + k] : 0);
        }
        JF_TRY(msm_multi_device(table, mb, (uint32_t*)(d_xyzz + b0 * xz), st));
    }
    if (d_res) JF_TRY(g1_batch_to_affine(srs->curve, (const uint32_t*)d_xyzz, batch, (uint32_t*)d_res, rec, st));
    return JFHIP_OK;
}

int32_t jfhip_msm_batch_partial_device(const jfhip_srs* srs, const size_t* offsets, const void* const* d_scalars,
                                       const size_t* lens, size_t batch, void* d_out_xyzz, jfhip_stream* stream) {
    MarkerRange mr("batch commit %zu polynomials (partial sums of a point range)", batch);
    if (!srs) {
        set_error("null SRS handle");
        return JFHIP_ERR_INVALID;
    }
    if (batch == 0) return JFHIP_OK;
    if (!d_scalars || !lens || !d_out_xyzz) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    for (size_t b = 0; b < batch; b++) {
        JF_TRY(check_msm_range(srs, offsets ? offsets[b] : 0, lens[b]));
        if (lens[b] && !d_scalars[b]) {
            set_error("null scalar pointer for batch entry %zu", b);
            return JFHIP_ERR_INVALID;
        }
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    const size_t xz = 4 * (size_t)fq_bytes(srs->curve);
    if (msm_batch_fusable(srs, lens, batch))
        return msm_batch_fused_locked(srs, offsets, d_scalars, lens, batch, nullptr, 0, (uint8_t*)d_out_xyzz, st);
    for (size_t b = 0; b < batch; b++) {
        const size_t off = offsets ? offsets[b] : 0;
        JF_TRY(msm_partial_device(srs_route(srs, off + lens[b], st), off, (const uint32_t*)d_scalars[b], lens[b],
                                  (uint32_t*)((uint8_t*)d_out_xyzz + b * xz), st));
    }
    return JFHIP_OK;
}

int32_t jfhip_msm_batch_device(const jfhip_srs* srs, const void* const* d_scalars, const size_t* lens, size_t batch,
                               void* d_out, jfhip_stream* stream) {
    MarkerRange mr("batch commit %zu polynomials (device scalars)", batch);
    if (!srs) {
        set_error("null SRS handle");
        return JFHIP_ERR_INVALID;
    }
    if (batch == 0) return JFHIP_OK;
    if (!d_scalars || !lens || !d_out) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    for (size_t b = 0; b < batch; b++) {
        JF_TRY(check_msm_range(srs, 0, lens[b]));
        if (lens[b] && !d_scalars[b]) {
            set_error("null scalar pointer for batch entry %zu", b);
            return JFHIP_ERR_INVALID;
        }
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    const size_t rec = (2 * (size_t)fq_bytes(srs->curve) + 4 + 15) & ~(size_t)15;
    if (msm_batch_fusable(srs, lens, batch))
        return msm_batch_fused_locked(srs, nullptr, d_scalars, lens, batch, (uint8_t*)d_out, rec, nullptr, st);
    return msm_batch_locked(srs, lens, batch, (uint8_t*)d_out, rec, st,
                            [&](size_t b, jfhip_stream*, const void** out) -> int32_t {
                                *out = d_scalars[b];
                                return JFHIP_OK;
                            });
}

int32_t jfhip_msm_small(int32_t curve_id, const uint64_t* points_xy, const uint64_t* scalars, size_t n, void* out_xy,
                        uint8_t* out_infinity, jfhip_stream* stream) {
    if (!out_xy || !out_infinity || (n && (!points_xy || !scalars))) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_TRY(check_curve(curve_id));
    if (n == 0) {
        memset(out_xy, 0, 2 * (size_t)fq_bytes(curve_id));
        *out_infinity = 1;
        return JFHIP_OK;
    }
    // bases change from call to call (the verifier's commitments), so the window table is built for this one product
    // and dropped: the same kernels and the same exactness as the SRS path
    jfhip_srs* tmp = nullptr;
    JF_TRY(jfhip_srs_load(curve_id, points_xy, n, 2 * (size_t)fq_bytes(curve_id), 0, &tmp));
    int32_t rc = jfhip_msm(tmp, 0, scalars, n, out_xy, out_infinity, stream);
    jfhip_srs_free(tmp);
    return rc;
}

int32_t jfhip_msm_batch(const jfhip_srs* srs, const uint64_t* const* scalars, const size_t* lens, size_t batch,
                        void* out_xy, uint8_t* out_inf, jfhip_stream* stream) {
    MarkerRange mr("batch commit %zu polynomials", batch);
    if (!srs) {
        set_error("null SRS handle");
        return JFHIP_ERR_INVALID;
    }
    if (batch == 0) return JFHIP_OK;  // gpu_batch_commit: empty input -> Ok(vec![]) (mod.rs:706-708)
    if (!scalars || !lens || !out_xy || !out_inf) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    size_t maxlen = 0;
    for (size_t b = 0; b < batch; b++) {
        JF_TRY(check_msm_range(srs, 0, lens[b]));
        if (lens[b] && !scalars[b]) {
            set_error("null scalar pointer for batch entry %zu", b);
            return JFHIP_ERR_INVALID;
        }
        if (lens[b] > maxlen) maxlen = lens[b];
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    const size_t pt = 2 * (size_t)fq_bytes(srs->curve);
    const size_t rec = (pt + 4 + 15) & ~(size_t)15;
    JF_TRY(st->ntt_io.ensure(batch * rec));  // result records
    uint8_t* d_res = st->ntt_io.as<uint8_t>();
    JF_TRY(msm_batch_locked(srs, lens, batch, d_res, rec, st,
                            [&](size_t b, jfhip_stream* ls, const void** out) -> int32_t {
                                // each lane stages its scalars in its own buffer, on its own stream
                                JF_TRY(ls->scalars.ensure((maxlen ? maxlen : 1) * 32));
                                if (lens[b])
                                    JF_HIP(hipMemcpyAsync(ls->scalars.p, scalars[b], lens[b] * 32,
                                                          hipMemcpyHostToDevice, ls->s));
                                *out = ls->scalars.p;
                                return JFHIP_OK;
                            }));
    std::vector<uint8_t> host(batch * rec);
    JF_HIP(hipMemcpyAsync(host.data(), d_res, batch * rec, hipMemcpyDeviceToHost, st->s));
    JF_HIP(hipStreamSynchronize(st->s));
    for (size_t b = 0; b < batch; b++) {
        memcpy((uint8_t*)out_xy + b * pt, host.data() + b * rec, pt);
        uint32_t inf;
        memcpy(&inf, host.data() + b * rec + pt, 4);
        out_inf[b] = inf ? 1 : 0;
    }
    return JFHIP_OK;
}

int32_t jfhip_msm(const jfhip_srs* srs, size_t offset, const uint64_t* scalars, size_t n, void* out_xy,
                  uint8_t* out_inf, jfhip_stream* stream) {
    MarkerRange mr("Committing to polynomial of degree %zu", n ? n - 1 : 0);
    JF_TRY(check_msm_range(srs, offset, n));
    if ((n && !scalars) || !out_xy || !out_inf) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    const size_t pt = 2 * (size_t)fq_bytes(srs->curve);
    // Host scalars: the upload (32 B per term over PCIe: 1.0 ms for 2^20 terms from pageable memory) is a large part of
    // the call.  From 2^20 terms on the MSM runs as 2 (4 from 2^22 terms) partial sums over point ranges on the lane
    // streams, so the upload of one range overlaps the kernels of the previous one; the partial sums are added and
    // normalised once.  Measured (scripts/host_msm_bench.py, BN254): 2^20 terms 3.08 -> 2.94 ms, 2^22 10.9 -> 8.7,
    // 2^24 39.1 -> 28.0; below 2^20 the duplicated tails cost more than the overlap gains.
    // JFHIP_MSM_UPLOAD_CHUNKS=1 keeps the single upload.
    static const int env_chunks = getenv("JFHIP_MSM_UPLOAD_CHUNKS") ? atoi(getenv("JFHIP_MSM_UPLOAD_CHUNKS")) : 0;
    size_t chunks = n >= ((size_t)1 << 22) ? 4 : n >= ((size_t)1 << 20) ? 2 : 1;
    if (env_chunks >= 1 && env_chunks <= 8) chunks = (size_t)env_chunks;
    if (chunks > n) chunks = n ? n : 1;
    const uint8_t* d_result = nullptr;
    if (chunks > 1) {
        const size_t xz = 2 * pt;
        JF_TRY(st->ntt_io.ensure(chunks * xz + pt + 16));
        uint8_t* d_part = st->ntt_io.as<uint8_t>();
        std::vector<size_t> lens(chunks), offs(chunks);
        const size_t per = (n + chunks - 1) / chunks;
        for (size_t b = 0; b < chunks; b++) {
            const size_t lo = b * per < n ? b * per : n, hi = (b + 1) * per < n ? (b + 1) * per : n;
            offs[b] = offset + lo;
            lens[b] = hi - lo;
        }
        JF_TRY(msm_batch_locked(
            srs, lens.data(), chunks, d_part, xz, st,
            [&](size_t b, jfhip_stream* ls, const void** out) -> int32_t {
                JF_TRY(ls->scalars.ensure((per ? per : 1) * 32));
                if (lens[b])
                    JF_HIP(hipMemcpyAsync(ls->scalars.p, scalars + (offs[b] - offset) * 4, lens[b] * 32,
                                          hipMemcpyHostToDevice, ls->s));
                *out = ls->scalars.p;
                return JFHIP_OK;
            },
            nullptr, offs.data()));
        JF_TRY(g1_sum_to_affine(srs->curve, (const uint32_t*)d_part, chunks, (uint32_t*)(d_part + chunks * xz), st));
        d_result = d_part + chunks * xz;
    } else {
        JF_TRY(st->scalars.ensure((n ? n : 1) * 32));
        JF_TRY(st->ntt_io.ensure(pt + 16));
        if (n) JF_HIP(hipMemcpyAsync(st->scalars.p, scalars, n * 32, hipMemcpyHostToDevice, st->s));
        JF_TRY(msm_device_locked(srs, offset, st->scalars.p, n, st->ntt_io.p, st));
        d_result = st->ntt_io.as<uint8_t>();
    }
    uint8_t host[160];
    JF_HIP(hipMemcpyAsync(host, d_result, pt + 4, hipMemcpyDeviceToHost, st->s));
    JF_HIP(hipStreamSynchronize(st->s));
    memcpy(out_xy, host, pt);
    uint32_t inf;
    memcpy(&inf, host + pt, 4);
    *out_inf = inf ? 1 : 0;
    return JFHIP_OK;
}

int32_t jfhip_kzg_open(const jfhip_srs* srs, const uint64_t* coeffs, size_t n, const uint64_t* point, void* out_xy,
                       uint8_t* out_inf, uint64_t* out_eval, jfhip_stream* stream) {
    MarkerRange mr("Opening polynomial of degree %zu", n ? n - 1 : 0);
    if (!srs || !point || !out_xy || !out_inf || !out_eval || (n && !coeffs)) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    const size_t pt = 2 * (size_t)fq_bytes(srs->curve);
    if (n == 0) {
        memset(out_xy, 0, pt);
        *out_inf = 1;
        memset(out_eval, 0, 32);
        return JFHIP_OK;
    }
    JF_TRY(check_msm_range(srs, 0, n - 1));
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    JF_TRY(st->poly_a.ensure(n * 32 + 32));
    JF_TRY(st->poly_c.ensure(n * 32));
    JF_TRY(st->ntt_io.ensure(pt + 16));
    uint8_t* d_coeffs = st->poly_a.as<uint8_t>() + 32;
    JF_HIP(hipMemcpyAsync(st->poly_a.p, point, 32, hipMemcpyHostToDevice, st->s));
    JF_HIP(hipMemcpyAsync(d_coeffs, coeffs, n * 32, hipMemcpyHostToDevice, st->s));
    JF_TRY(open_witness_device(srs->curve, (const uint32_t*)d_coeffs, n, st->poly_a.as<uint32_t>(),
                               st->poly_c.as<uint32_t>(), nullptr, st));
    // suffix[0] = f(z); suffix[1..n) = witness coefficients
    JF_TRY(msm_device_locked(srs, 0, st->poly_c.as<uint8_t>() + 32, n - 1, st->ntt_io.p, st));
    uint8_t host[160];
    JF_HIP(hipMemcpyAsync(host, st->ntt_io.p, pt + 4, hipMemcpyDeviceToHost, st->s));
    JF_HIP(hipMemcpyAsync(out_eval, st->poly_c.p, 32, hipMemcpyDeviceToHost, st->s));
    JF_HIP(hipStreamSynchronize(st->s));
    memcpy(out_xy, host, pt);
    uint32_t inf;
    memcpy(&inf, host + pt, 4);
    *out_inf = inf ? 1 : 0;
    return JFHIP_OK;
}

// ---- multi_point_open (pcs/src/univariate_kzg/mod.rs:416-468) -----------------------------------------------------------
// evals[i] = f(z_i); witness = f / prod_i (X - z_i) with the remainder discarded (ark's DensePolynomial division);
// proof = commit(witness).  The quotient by the product is k successive synthetic divisions, each dropping its remainder:
// floor(floor(f / A) / B) = floor(f / (A B)) for polynomials.  Everything on the device; k is small by the reference's own
// warning ("don't use it when points.len() is large").
int32_t jfhip_kzg_multi_point_open(const jfhip_srs* srs, const uint64_t* coeffs, size_t n, const uint64_t* points,
                                   size_t num_points, void* out_xy, uint8_t* out_inf, uint64_t* out_evals,
                                   jfhip_stream* stream) {
    MarkerRange mr("Opening polynomial of degree %zu at %zu points", n ? n - 1 : 0, num_points);
    if (num_points == 0) {
        set_error("no point to evaluate and open");   // mod.rs:421-425
        return JFHIP_ERR_INVALID;
    }
    if (!srs || !points || !out_xy || !out_inf || !out_evals || (n && !coeffs)) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    const size_t pt = 2 * (size_t)fq_bytes(srs->curve);
    const size_t k = num_points;
    if (n == 0) {
        memset(out_xy, 0, pt);
        *out_inf = 1;
        memset(out_evals, 0, k * 32);
        return JFHIP_OK;
    }
    const size_t wlen = n > k ? n - k : 0;   // coefficients of the witness polynomial
    JF_TRY(check_msm_range(srs, 0, wlen));
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    JF_TRY(st->poly_a.ensure(n * 32));
    JF_TRY(st->poly_c.ensure(n * 32));
    JF_TRY(st->quot_consts.ensure(2 * k * 32));   // the points, then the evaluations
    JF_TRY(st->ntt_io.ensure(pt + 16));
    uint32_t* d_pts = st->quot_consts.as<uint32_t>();
    uint32_t* d_evals = d_pts + k * 8;
    JF_HIP(hipMemcpyAsync(d_pts, points, k * 32, hipMemcpyHostToDevice, st->s));
    JF_HIP(hipMemcpyAsync(st->poly_a.p, coeffs, n * 32, hipMemcpyHostToDevice, st->s));
    const uint32_t* d_f = st->poly_a.as<uint32_t>();
    // polynomial.evaluate(point) for every point (:433-436)
    for (size_t i = 0; i < k; i++)
        JF_TRY(poly_eval_batch_device(srs->curve, &d_f, &n, 1, d_pts + i * 8, d_evals + i * 8, st));
    // the division chain, ping-ponging between the two buffers: a step writes [remainder, quotient...]
    const uint32_t* cur = d_f;
    size_t len = n;
    uint32_t* bufs[2] = {st->poly_c.as<uint32_t>(), st->poly_a.as<uint32_t>()};
    for (size_t i = 0; i < k && len > 0; i++) {
        uint32_t* dst = bufs[i & 1];
        JF_TRY(open_witness_device(srs->curve, cur, len, d_pts + i * 8, dst, nullptr, st));
        cur = dst + 8;
        len -= 1;
    }
    JF_TRY(msm_device_locked(srs, 0, cur, len, st->ntt_io.p, st));
    uint8_t host[160];
    JF_HIP(hipMemcpyAsync(host, st->ntt_io.p, pt + 4, hipMemcpyDeviceToHost, st->s));
    JF_HIP(hipMemcpyAsync(out_evals, d_evals, k * 32, hipMemcpyDeviceToHost, st->s));
    JF_HIP(hipStreamSynchronize(st->s));
    memcpy(out_xy, host, pt);
    uint32_t inf;
    memcpy(&inf, host + pt, 4);
    *out_inf = inf ? 1 : 0;
    return JFHIP_OK;
}

// ---- FK23 multi-open at roots of unity ----------------------------------------------

int32_t jfhip_kzg_multi_open_rou(const jfhip_srs* srs, const uint64_t* coeffs, size_t n, uint32_t log_domain,
                                 size_t num_points, void* out_xy, uint8_t* out_inf, jfhip_stream* stream) {
    if (!srs || (n && !coeffs) || (num_points && (!out_xy || !out_inf))) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    const uint32_t adicity = srs->curve == JFHIP_BN254 ? 28u : 32u;
    if (log_domain > adicity || log_domain > 26) {
        set_error("multi_open_rou: domain of 2^%u points unsupported", log_domain);
        return JFHIP_ERR_INVALID;
    }
    const size_t dom = (size_t)1 << log_domain;
    if (num_points > dom) {
        set_error("multi_open_rou: %zu points exceed the domain size %zu", num_points, dom);
        return JFHIP_ERR_INVALID;
    }
    // the h-polynomial has n - 1 coefficients: it must fit the domain (batch_evaluate_rou, pcs/src/poly.rs:143-150)
    if (n > 1 && n - 1 > dom) {
        set_error("Polynomial with %zu num_of_coeffs can't be evaluated on a smaller domain with size %zu", n - 1, dom);
        return JFHIP_ERR_INVALID;
    }
    if (num_points == 0) return JFHIP_OK;
    const size_t pt = 2 * (size_t)fq_bytes(srs->curve);
    if (n <= 1) {   // constant polynomial: every quotient (f - f(z)) / (X - z) is zero
        memset(out_xy, 0, num_points * pt);
        memset(out_inf, 1, num_points);
        return JFHIP_OK;
    }
    JF_TRY(check_msm_range(srs, 0, n - 1));
    uint32_t log_M = 1;
    while (((size_t)1 << log_M) < 2 * (n - 1)) log_M++;
    if (log_M > adicity || log_M > 27) {
        set_error("multi_open_rou: polynomial too long");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    const size_t rec = (pt + 4 + 15) & ~(size_t)15;
    JF_TRY(st->poly_a.ensure(n * 32));
    JF_TRY(st->ntt_io.ensure(num_points * rec));
    JF_HIP(hipMemcpyAsync(st->poly_a.p, coeffs, n * 32, hipMemcpyHostToDevice, st->s));
    JF_TRY(multi_open_rou_device(srs, st->poly_a.as<uint32_t>(), n, log_domain, num_points, st->ntt_io.as<uint32_t>(), rec,
                                 log_M, st));
    std::vector<uint8_t> host(num_points * rec);
    JF_HIP(hipMemcpyAsync(host.data(), st->ntt_io.p, num_points * rec, hipMemcpyDeviceToHost, st->s));
    JF_HIP(hipStreamSynchronize(st->s));
    for (size_t i = 0; i < num_points; i++) {
        memcpy((uint8_t*)out_xy + i * pt, host.data() + i * rec, pt);
        uint32_t inf;
        memcpy(&inf, host.data() + i * rec + pt, 4);
        out_inf[i] = inf ? 1 : 0;
    }
    return JFHIP_OK;
}

// ---- multilinear KZG (the MSM sites of pcs/src/multilinear_kzg/mod.rs) --------------

int32_t jfhip_mle_fold_device(int32_t curve_id, const void* d_f, size_t half, const uint64_t* point, void* d_q, void* d_r,
                              jfhip_stream* stream) {
    JF_TRY(check_curve(curve_id));
    if (!point || (half && (!d_f || !d_q || !d_r))) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    JF_TRY(st->quot_consts.ensure(32));
    JF_HIP(hipMemcpyAsync(st->quot_consts.p, point, 32, hipMemcpyHostToDevice, st->s));
    return mle_fold_device(curve_id, (const uint32_t*)d_f, half, st->quot_consts.as<uint32_t>(), (uint32_t*)d_q,
                           (uint32_t*)d_r, st);
}

int32_t jfhip_mle_open(const jfhip_srs* const* levels, size_t nv, const uint64_t* evals, const uint64_t* point,
                       void* out_xy, uint8_t* out_inf, uint64_t* out_eval, jfhip_stream* stream) {
    if (!evals || !out_eval || (nv && (!levels || !point || !out_xy || !out_inf))) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    if (nv > 30) {
        set_error("mle open: %zu variables unsupported (max 30)", nv);
        return JFHIP_ERR_INVALID;
    }
    if (nv == 0) {
        memcpy(out_eval, evals, 32);
        return JFHIP_OK;
    }
    const int32_t curve = levels[0] ? levels[0]->curve : -1;
    for (size_t i = 0; i < nv; i++) {
        if (!levels[i] || levels[i]->curve != curve) {
            set_error("mle open: null or mixed-curve SRS level %zu", i);
            return JFHIP_ERR_INVALID;
        }
        // round i commits 2^(nv - 1 - i) values against powers_of_g[ignored + i].evals (mod.rs:322-350)
        JF_TRY(check_msm_range(levels[i], 0, (size_t)1 << (nv - 1 - i)));
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    const size_t n = (size_t)1 << nv;
    const size_t pt = 2 * (size_t)fq_bytes(curve);
    const size_t rec = (pt + 4 + 15) & ~(size_t)15;
    JF_TRY(st->poly_a.ensure(n * 32));          // f
    JF_TRY(st->poly_b.ensure(n * 16 + 32));      // r (next f), ping-pong with f
    JF_TRY(st->poly_c.ensure(n * 32));           // q of every round, back to back (n - 1 values in all)
    JF_TRY(st->quot_consts.ensure(nv * 32));
    JF_TRY(st->ntt_io.ensure(nv * rec));
    JF_HIP(hipMemcpyAsync(st->poly_a.p, evals, n * 32, hipMemcpyHostToDevice, st->s));
    JF_HIP(hipMemcpyAsync(st->quot_consts.p, point, nv * 32, hipMemcpyHostToDevice, st->s));
    uint32_t* f = st->poly_a.as<uint32_t>();
    uint32_t* r = st->poly_b.as<uint32_t>();
    // all folds first (round i + 1 needs only r of round i), every q kept; then the nv commitments as one batch: the
    // short MSMs of the late rounds are latency-bound and overlap on the library's lane streams
    std::vector<size_t> lens(nv), q_at(nv);
    size_t at = 0;
    for (size_t i = 0; i < nv; i++) {
        const size_t half = (size_t)1 << (nv - 1 - i);
        lens[i] = half;
        q_at[i] = at;
        JF_TRY(mle_fold_device(curve, f, half, st->quot_consts.as<uint32_t>() + i * 8,
                               st->poly_c.as<uint32_t>() + at * 8, r, st));
        at += half;
        uint32_t* t = f;
        f = r;
        r = t;
    }
    const uint8_t* q_base = st->poly_c.as<uint8_t>();
    JF_TRY(msm_batch_locked(
        levels[0], lens.data(), nv, st->ntt_io.as<uint8_t>(), rec, st,
        [&](size_t b, jfhip_stream*, const void** d) {
            *d = q_base + q_at[b] * 32;
            return (int32_t)JFHIP_OK;
        },
        [&](size_t b) { return levels[b]; }));
    std::vector<uint8_t> host(nv * rec);
    JF_HIP(hipMemcpyAsync(host.data(), st->ntt_io.p, nv * rec, hipMemcpyDeviceToHost, st->s));
    JF_HIP(hipMemcpyAsync(out_eval, f, 32, hipMemcpyDeviceToHost, st->s));   // the table folded to one value: f(point)
    JF_HIP(hipStreamSynchronize(st->s));
    for (size_t i = 0; i < nv; i++) {
        memcpy((uint8_t*)out_xy + i * pt, host.data() + i * rec, pt);
        uint32_t inf;
        memcpy(&inf, host.data() + i * rec + pt, 4);
        out_inf[i] = inf ? 1 : 0;
    }
    return JFHIP_OK;
}

// ---- NTT ----------------------------------------------------------------------------------------

int32_t jfhip_ntt_device(int32_t curve_id, void* d_data, uint32_t log_n, size_t batch, int32_t inverse,
                         const uint64_t* coset, size_t in_len, jfhip_stream* stream) {
    MarkerRange mr("%s%sFFT of %zu polynomials over a domain of size 2^%u", coset ? "coset " : "", inverse ? "i" : "", batch,
                   log_n);
    JF_TRY(check_curve(curve_id));
    if (batch && !d_data) {
        set_error("null data pointer");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    return ntt_device(curve_id, (uint32_t*)d_data, log_n, batch, inverse, coset, in_len, st);
}

int32_t jfhip_ntt(int32_t curve_id, uint64_t* data, uint32_t log_n, size_t batch, int32_t inverse,
                  const uint64_t* coset, size_t in_len, jfhip_stream* stream) {
    MarkerRange mr("%s%sFFT of %zu polynomials over a domain of size 2^%u (host data)", coset ? "coset " : "",
                   inverse ? "i" : "", batch, log_n);
    JF_TRY(check_curve(curve_id));
    if (batch == 0) return JFHIP_OK;
    if (!data) {
        set_error("null data pointer");
        return JFHIP_ERR_INVALID;
    }
    if (log_n > 30) {
        set_error("ntt: log_n %u too large", log_n);
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    const size_t n = (size_t)1 << log_n;
    if (in_len == 0 || in_len > n) in_len = n;
    const size_t bytes = batch * n * 32;
    JF_TRY(st->ntt_io.ensure(bytes));
    if (in_len == n) {
        JF_HIP(hipMemcpyAsync(st->ntt_io.p, data, bytes, hipMemcpyHostToDevice, st->s));
    } else {
        // only the first in_len elements of each polynomial cross PCIe
        JF_HIP(hipMemcpy2DAsync(st->ntt_io.p, n * 32, data, n * 32, in_len * 32, batch, hipMemcpyHostToDevice,
                                st->s));
        if (log_n == 0) in_len = n;
    }
    JF_TRY(ntt_device(curve_id, st->ntt_io.as<uint32_t>(), log_n, batch, inverse, coset, in_len, st));
    JF_HIP(hipMemcpyAsync(data, st->ntt_io.p, bytes, hipMemcpyDeviceToHost, st->s));
    JF_HIP(hipStreamSynchronize(st->s));
    return JFHIP_OK;
}

int32_t jfhip_plonk_quotient_turbo_device(int32_t curve_id, uint32_t log_n, uint32_t log_m, const void* d_sel,
                                          const void* d_sig, const void* d_wires, const void* d_zperm,
                                          const void* d_pi, const uint64_t* k_mont, const uint64_t* ch, void* d_quot,
                                          jfhip_stream* stream) {
    JF_TRY(check_curve(curve_id));
    if (!d_sel || !d_sig || !d_wires || !d_zperm || !d_pi || !k_mont || !ch || !d_quot) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    return plonk_quotient_turbo_device(curve_id, log_n, log_m, (const uint32_t*)d_sel, (const uint32_t*)d_sig,
                                       (const uint32_t*)d_wires, (const uint32_t*)d_zperm, (const uint32_t*)d_pi,
                                       k_mont, ch, (uint32_t*)d_quot, st);
}

int32_t jfhip_plonk_quotient_ultra_device(int32_t curve_id, uint32_t log_n, uint32_t log_m, const void* d_sel,
                                          const void* d_sig, const void* d_wires, const void* d_zperm,
                                          const void* d_pi, const void* d_plookup, const uint64_t* k_mont,
                                          const uint64_t* ch, void* d_quot, jfhip_stream* stream) {
    JF_TRY(check_curve(curve_id));
    if (!d_sel || !d_sig || !d_wires || !d_zperm || !d_pi || !d_plookup || !k_mont || !ch || !d_quot) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    return plonk_quotient_ultra_device(curve_id, log_n, log_m, (const uint32_t*)d_sel, (const uint32_t*)d_sig,
                                       (const uint32_t*)d_wires, (const uint32_t*)d_zperm, (const uint32_t*)d_pi,
                                       (const uint32_t*)d_plookup, k_mont, ch, (uint32_t*)d_quot, st);
}

int32_t jfhip_plonk_quotient_coset_device(int32_t curve_id, uint32_t num_wire_types, uint32_t log_n, uint32_t log_m,
                                          const uint64_t* coset_offset_mont, const void* d_sel, const void* d_sig,
                                          const void* d_wires, const void* d_zperm, const void* d_pi,
                                          const void* d_plookup, const uint64_t* k_mont, const uint64_t* ch,
                                          void* d_quot, jfhip_stream* stream) {
    JF_TRY(check_curve(curve_id));
    if (num_wire_types != 5 && num_wire_types != 6) {
        set_error("quotient: %u wire types (5 = TurboPlonk, 6 = UltraPlonk)", num_wire_types);
        return JFHIP_ERR_INVALID;
    }
    if (!coset_offset_mont || !d_sel || !d_sig || !d_wires || !d_zperm || !d_pi || !k_mont || !ch || !d_quot ||
        (num_wire_types == 6 && !d_plookup)) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    return plonk_quotient_coset_device(curve_id, (int)num_wire_types, log_n, log_m, coset_offset_mont,
                                       (const uint32_t*)d_sel, (const uint32_t*)d_sig, (const uint32_t*)d_wires,
                                       (const uint32_t*)d_zperm, (const uint32_t*)d_pi, (const uint32_t*)d_plookup,
                                       k_mont, ch, (uint32_t*)d_quot, st);
}

int32_t jfhip_plonk_quotient_masked_device(int32_t curve_id, uint32_t num_wire_types, uint32_t log_n, uint32_t log_m,
                                           const uint64_t* coset_offset_mont, const void* d_sel, const void* d_sig,
                                           const void* d_wires, const void* d_zperm, const void* d_pi,
                                           const void* d_plookup, const uint64_t* k_mont, const uint64_t* ch,
                                           uint32_t zero_selector_mask, void* d_quot, jfhip_stream* stream) {
    JF_TRY(check_curve(curve_id));
    if (num_wire_types != 5 && num_wire_types != 6) {
        set_error("quotient: %u wire types (5 = TurboPlonk, 6 = UltraPlonk)", num_wire_types);
        return JFHIP_ERR_INVALID;
    }
    const uint32_t nsel = num_wire_types == 5 ? 13u : 14u;
    if (zero_selector_mask >> nsel) {
        set_error("quotient: zero_selector_mask 0x%x names selectors beyond the %u of this circuit type",
                  zero_selector_mask, nsel);
        return JFHIP_ERR_INVALID;
    }
    const bool all_zero = zero_selector_mask == ((1u << nsel) - 1u);
    if (!coset_offset_mont || (!d_sel && !all_zero) || !d_sig || !d_wires || !d_zperm || !k_mont || !ch || !d_quot ||
        (num_wire_types == 6 && !d_plookup)) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    return plonk_quotient_coset_device(curve_id, (int)num_wire_types, log_n, log_m, coset_offset_mont,
                                       (const uint32_t*)d_sel, (const uint32_t*)d_sig, (const uint32_t*)d_wires,
                                       (const uint32_t*)d_zperm, (const uint32_t*)d_pi, (const uint32_t*)d_plookup,
                                       k_mont, ch, (uint32_t*)d_quot, st, zero_selector_mask);
}

int32_t jfhip_plonk_perm_product_device(int32_t curve_id, uint32_t log_n, uint32_t num_wire_types,
                                        const void* d_wire_vals, const void* d_id_perm, const void* d_sigma_vals,
                                        const uint64_t* beta, const uint64_t* gamma, void* d_out,
                                        jfhip_stream* stream) {
    JF_TRY(check_curve(curve_id));
    if (!d_wire_vals || !d_id_perm || !d_sigma_vals || !beta || !gamma || !d_out) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    return plonk_perm_product_device(curve_id, log_n, num_wire_types, (const uint32_t*)d_wire_vals,
                                     (const uint32_t*)d_id_perm, (const uint32_t*)d_sigma_vals, beta, gamma,
                                     (uint32_t*)d_out, st);
}

int32_t jfhip_plonk_lookup_product_device(int32_t curve_id, uint32_t log_n, const void* d_lookup_wire,
                                          const void* d_table, const void* d_sorted, const uint64_t* beta,
                                          const uint64_t* gamma, void* d_out, jfhip_stream* stream) {
    JF_TRY(check_curve(curve_id));
    if (!d_lookup_wire || !d_table || !d_sorted || !beta || !gamma || !d_out) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    return plonk_lookup_product_device(curve_id, log_n, (const uint32_t*)d_lookup_wire, (const uint32_t*)d_table,
                                       (const uint32_t*)d_sorted, beta, gamma, (uint32_t*)d_out, st);
}

int32_t jfhip_plonk_lookup_merge_device(int32_t curve_id, uint32_t log_n, const void* d_wire_vals,
                                        const void* d_table_vals, const uint64_t* tau, void* d_merged_table,
                                        void* d_merged_lookup, jfhip_stream* stream) {
    JF_TRY(check_curve(curve_id));
    if (!d_wire_vals || !d_table_vals || !tau || !d_merged_table || !d_merged_lookup) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    return plonk_lookup_merge_device(curve_id, log_n, (const uint32_t*)d_wire_vals, (const uint32_t*)d_table_vals, tau,
                                     (uint32_t*)d_merged_table, (uint32_t*)d_merged_lookup, st);
}

int32_t jfhip_plonk_lookup_sorted_device(const void* d_table, size_t n_table, const void* d_lookups, size_t n_lookups,
                                         void* d_sorted, jfhip_stream* stream) {
    if (!d_table || !d_sorted || (n_lookups && !d_lookups)) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    return plonk_lookup_sorted_device((const uint32_t*)d_table, n_table, (const uint32_t*)d_lookups, n_lookups,
                                      (uint32_t*)d_sorted, st);
}

int32_t jfhip_poly_divide_linear_device(int32_t curve_id, const void* d_coeffs, size_t n, const uint64_t* point,
                                        void* d_out, jfhip_stream* stream) {
    JF_TRY(check_curve(curve_id));
    if (n == 0) return JFHIP_OK;
    if (!d_coeffs || !point || !d_out) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    JF_TRY(st->poly_a.ensure(64));
    JF_HIP(hipMemcpyAsync(st->poly_a.p, point, 32, hipMemcpyHostToDevice, st->s));
    return open_witness_device(curve_id, (const uint32_t*)d_coeffs, n, st->poly_a.as<uint32_t>(), (uint32_t*)d_out,
                               nullptr, st);
}

int32_t jfhip_poly_divide_linear_noalloc_device(int32_t curve_id, const void* d_coeffs, size_t n, const uint64_t* point,
                                                void* d_out, jfhip_stream* stream) {
    JF_TRY(check_curve(curve_id));
    if (!stream) {
        // NULL would mean the per-thread default stream, which the library creates on first use: an allocation
        set_error("no-alloc divide: pass a stream made at set-up (jfhip_stream_create / jfhip_stream_wrap)");
        return JFHIP_ERR_INVALID;
    }
    if (n == 0) return JFHIP_OK;
    if (!d_coeffs || !point || !d_out) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(stream->mu);
    return poly_divide_linear_noalloc(curve_id, (const uint32_t*)d_coeffs, n, point, (uint32_t*)d_out, stream->s);
}

int32_t jfhip_poly_eval_batch_device(int32_t curve_id, const void* const* d_polys, const size_t* lens, size_t count,
                                     const uint64_t* point, void* d_out, jfhip_stream* stream) {
    JF_TRY(check_curve(curve_id));
    if (count == 0) return JFHIP_OK;
    if (!d_polys || !lens || !point || !d_out) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    JF_TRY(st->poly_a.ensure(64));
    JF_HIP(hipMemcpyAsync(st->poly_a.p, point, 32, hipMemcpyHostToDevice, st->s));
    return poly_eval_batch_device(curve_id, (const uint32_t* const*)d_polys, lens, count, st->poly_a.as<uint32_t>(),
                                  (uint32_t*)d_out, st);
}

int32_t jfhip_poly_lincomb_device(int32_t curve_id, const void* const* d_polys, const size_t* lens, size_t count,
                                  const uint64_t* coeffs, void* d_out, size_t n, jfhip_stream* stream) {
    JF_TRY(check_curve(curve_id));
    if (!d_polys || !lens || !coeffs || !d_out) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    return poly_lincomb_device(curve_id, (const uint32_t* const*)d_polys, lens, count, coeffs, (uint32_t*)d_out, n, st);
}

int32_t jfhip_witness_gather_device(const void* d_witness, size_t num_vars, const uint32_t* d_wire_variables,
                                    size_t count, void* d_out, int32_t check, jfhip_stream* stream) {
    if (count == 0) return JFHIP_OK;
    if (!d_witness || !d_wire_variables || !d_out || num_vars == 0 || num_vars >= ((size_t)1 << 32)) {
        set_error("witness gather: null argument or bad table size %zu", num_vars);
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, stream);
    std::lock_guard<std::mutex> lk(st->mu);
    JF_TRY(witness_gather_device((const uint32_t*)d_witness, num_vars, d_wire_variables, count, (uint32_t*)d_out, st));
    if (check) {
        uint32_t bad = 0;
        JF_HIP(hipMemcpyAsync(&bad, st->poly_c.as<uint32_t>() + 24, 4, hipMemcpyDeviceToHost, st->s));
        JF_HIP(hipStreamSynchronize(st->s));
        if (bad) {
            set_error("witness gather: a wire variable index is >= num_vars (%zu)", num_vars);
            return JFHIP_ERR_INVALID;
        }
    }
    return JFHIP_OK;
}

// ---- field / group test surface ---------------------------------------------------------------------

int32_t jfhip_field_op(int32_t field_id, int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    if (field_id < 0 || field_id > 3 || op < 0 || op > 13) {
        set_error("bad field id / op");
        return JFHIP_ERR_INVALID;
    }
    if (n == 0) return JFHIP_OK;
    if (!a || !out || ((op <= 2 || (op >= 7 && op <= 11)) && !b)) {
        set_error("null argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, nullptr);
    std::lock_guard<std::mutex> lk(st->mu);
    const size_t el = field_id == 3 ? 48 : 32;
    JF_TRY(st->poly_a.ensure(n * el));
    JF_TRY(st->poly_b.ensure(n * el));
    JF_TRY(st->poly_c.ensure(n * el));
    JF_HIP(hipMemcpyAsync(st->poly_a.p, a, n * el, hipMemcpyHostToDevice, st->s));
    if (b) JF_HIP(hipMemcpyAsync(st->poly_b.p, b, n * el, hipMemcpyHostToDevice, st->s));
    JF_TRY(field_op_device(field_id, op, st->poly_a.as<uint32_t>(), st->poly_b.as<uint32_t>(),
                           st->poly_c.as<uint32_t>(), n, st->s));
    JF_HIP(hipMemcpyAsync(out, st->poly_c.p, n * el, hipMemcpyDeviceToHost, st->s));
    JF_HIP(hipStreamSynchronize(st->s));
    return JFHIP_OK;
}

int32_t jfhip_g1_op(int32_t curve_id, int32_t op, const uint64_t* P, const uint64_t* Q, uint64_t* out) {
    JF_TRY(check_curve(curve_id));
    if (!P || !Q || !out || op < 0 || op > 4) {
        set_error("bad argument");
        return JFHIP_ERR_INVALID;
    }
    JF_STREAM(st, nullptr);
    std::lock_guard<std::mutex> lk(st->mu);
    const size_t pt = 2 * (size_t)fq_bytes(curve_id);
    JF_TRY(st->poly_a.ensure(3 * pt));
    uint8_t* d = st->poly_a.as<uint8_t>();
    JF_HIP(hipMemcpyAsync(d, P, pt, hipMemcpyHostToDevice, st->s));
    JF_HIP(hipMemcpyAsync(d + pt, Q, (op == 0 || op == 2) ? pt : 32, hipMemcpyHostToDevice, st->s));
    JF_TRY(g1_op_device(curve_id, op, (uint32_t*)d, (uint32_t*)(d + pt), (uint32_t*)(d + 2 * pt), st->s));
    JF_HIP(hipMemcpyAsync(out, d + 2 * pt, pt, hipMemcpyDeviceToHost, st->s));
    JF_HIP(hipStreamSynchronize(st->s));
    return JFHIP_OK;
}

int32_t jfhip_probe(int32_t kind, int32_t arg, double* out_rate) {
    if (!out_rate) return JFHIP_ERR_INVALID;
    return probe(kind, arg, out_rate);
}

int32_t jfhip_marker_push(const char* name) {
    const Roctx& r = roctx();
    if (r.push && name) r.push(name);
    return JFHIP_OK;
}

int32_t jfhip_marker_pop(void) {
    const Roctx& r = roctx();
    if (r.pop) r.pop();
    return JFHIP_OK;
}

int32_t jfhip_set_profiling(int32_t enabled) {
    g_profiling = enabled != 0;
    return JFHIP_OK;
}

int32_t jfhip_last_phase_ms(jfhip_stream* stream, double* phases, int32_t cap) {
    jfhip_stream* st = resolve_stream(stream);
    if (!st || !phases || !st->ev_ready) return 0;
    std::lock_guard<std::mutex> lk(st->mu);
    if (hipStreamSynchronize(st->s) != hipSuccess) return 0;
    int n = st->n_phases < cap ? st->n_phases : cap;
    for (int i = 0; i < n; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, st->ev[i], st->ev[i + 1]) != hipSuccess) ms = -1.f;
        phases[i] = ms;
    }
    return n;
}

const char* jfhip_phase_name(int32_t kind, int32_t idx) {
    static const char* msm[] = {"coarse_histogram", "bucket_sort", "bucket_accumulate", "slice_fixup",
                                "bucket_fold",      "sum_tree"};
    if (kind == 0) return (idx >= 0 && idx < 6) ? msm[idx] : "";
    static const char* ntt[] = {"pass0", "pass1", "pass2", "pass3"};
    return (idx >= 0 && idx < 4) ? ntt[idx] : "";
}

}  // extern "C"
