// context.cpp — device context: stream, workspace arena, per-domain twiddle cache; host circle-group helpers.
#include <cstring>
#include <cstdlib>
#include <string.h>

#include <cstdio>

#include <algorithm>
#include <mutex>

#include "host.h"

extern char** environ;  // (POSIX; scanned once for FRIEDA_* names nobody reads)

namespace frieda {

// ---- circle group on the host ------------------------------------------------------------------
static constexpr uint32_t IDX_MASK = 0x7fffffffu;

CPoint point_from_index(uint32_t index) {
    // CirclePointIndex::to_point: scalar multiple of the order-2^31 generator.  The 31 doublings of the generator are a table (the host
    // verifier evaluates a domain point per query and layer: src/proof.rs:98-100, ~400 calls per proof), so a call is one point addition
    // per set bit of the index.
    static const std::array<CPoint, 31> pow2 = [] {
        std::array<CPoint, 31> t{};
        CPoint cur{CIRCLE_GEN_X, CIRCLE_GEN_Y};
        for (auto& e : t) {
            e = cur;
            cur = cp_double(cur);
        }
        return t;
    }();
    CPoint res{1, 0};
    index &= IDX_MASK;
    for (uint32_t b = 0; index; b++, index >>= 1)
        if (index & 1u) res = cp_add(res, pow2[b]);
    return res;
}

Coset Coset::half_odds(uint32_t log_size) {
    // Coset::new(subgroup_gen(log_size + 2), log_size); subgroup_gen(k) = 2^(31-k)
    Coset c;
    c.initial = 1u << (31 - (log_size + 2));
    c.step = log_size == 0 ? 0u : (1u << (31 - log_size));
    c.log_size = log_size;
    return c;
}
Coset Coset::doubled() const { return Coset{(initial * 2u) & IDX_MASK, (step * 2u) & IDX_MASK, log_size - 1}; }
uint32_t Coset::index_at(uint32_t i) const { return (initial + (uint32_t)((uint64_t)step * i)) & IDX_MASK; }
CPoint Coset::at(uint32_t i) const { return point_from_index(index_at(i)); }

Coset line_coset(uint32_t n, uint32_t m) {
    Coset c = Coset::half_odds(n - 1);
    while (c.log_size > m) c = c.doubled();
    return c;
}

CodecShape codec_shape(size_t len) {
    // src/utils.rs:10-33.  F = ceil(8 len / 30); F' = 2^max(ceil(log2 F), 2) (the reference computes the
    // exponent in f64; identical for every F < 2^52); L = log2(F') - 2.
    CodecShape s;
    s.n_felts = (8 * len + 29) / 30;
    uint32_t e = 2;
    while (((size_t)1 << e) < s.n_felts) e++;
    s.n_padded = (size_t)1 << e;
    s.log_size = e - 2;
    return s;
}

// ---- kernel timer ------------------------------------------------------------------------------
struct KernelTimerImpl {
    struct Span {
        const char* name;
        double alg_bytes;
        hipEvent_t start, stop;
    };
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    hipEvent_t next() {
        if (used == pool.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            pool.push_back(e);
        }
        return pool[used++];
    }
    ~KernelTimerImpl() {
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

namespace k {
void timer_begin(KernelTimer* t, hipStream_t s, const char* name, double alg_bytes) {
    auto* ti = reinterpret_cast<KernelTimerImpl*>(t);
    KernelTimerImpl::Span sp{name, alg_bytes, ti->next(), ti->next()};
    (void)hipEventRecord(sp.start, s);
    ti->spans.push_back(sp);
}
void timer_end(KernelTimer* t, hipStream_t s) {
    auto* ti = reinterpret_cast<KernelTimerImpl*>(t);
    (void)hipEventRecord(ti->spans.back().stop, s);
}
}  // namespace k

k::Launch Ctx::launch() const {
    k::Launch l{stream, reinterpret_cast<k::KernelTimer*>(timer)};
    l.tune = &tuning;
    return l;
}

namespace k {
namespace {
struct KnobDesc {
    const char* name;
    long lo, hi;
    bool (*set)(Tuning&, long);  // false: a value inside [lo, hi] that the knob does not support (nothing is changed)
};
const KnobDesc KNOBS[] = {
    {"FRIEDA_T5_WIDE_LOG", 16, 24, [](Tuning& t, long v) { t.t5_wide_log = (uint32_t)v; return true; }},
    {"FRIEDA_T5_REG3_LOG", 0, 30, [](Tuning& t, long v) { t.t5_reg3_log = (uint32_t)v; return true; }},
    {"FRIEDA_T9_MAX_LOG", 8, 19, [](Tuning& t, long v) { t.t9_max_log = (uint32_t)v; return true; }},
    {"FRIEDA_TOP_MAX_LOG", 9, 11, [](Tuning& t, long v) { t.top_max_log = (uint32_t)v; return true; }},
    {"FRIEDA_NTT_CPW", 1, 4, [](Tuning& t, long v) {
         if (v == 4 && !t.lds_opt_in_ok) return false;  // four columns per workgroup need the 68 KB LDS opt-in this device refused
         t.ntt_cpw = (uint32_t)v;
         return true;
     }},
    {"FRIEDA_NTT_CPW_SMALL", 1, 4, [](Tuning& t, long v) { t.ntt_cpw_small = (uint32_t)v; return true; }},
    {"FRIEDA_NTT_REP", 0, 2, [](Tuning& t, long v) { t.ntt_rep = (uint32_t)v; return true; }},
    {"FRIEDA_NTT_NO_CP", 0, 1, [](Tuning& t, long v) { t.ntt_no_cp = v != 0; return true; }},
    {"FRIEDA_NTT_NO_PAD8", 0, 1, [](Tuning& t, long v) { t.ntt_no_pad8 = v != 0; return true; }},
    {"FRIEDA_NTT_TREE_REG_ONLY", 0, 1, [](Tuning& t, long v) { t.ntt_tree_reg_only = v != 0; return true; }},
    {"FRIEDA_NO_ENCODE_TREE_FUSION", 0, 1, [](Tuning& t, long v) { t.no_encode_tree_fusion = v != 0; return true; }},
    {"FRIEDA_ENCODE_TREE_FUSION_PROVE", 0, 1, [](Tuning& t, long v) { t.encode_tree_fusion_prove = v != 0; return true; }},
    {"FRIEDA_NO_SMALL_FUSED", 0, 1, [](Tuning& t, long v) { t.no_small_fused = v != 0; return true; }},
    {"FRIEDA_UNPACK_TILES", 1, 8, [](Tuning& t, long v) {
         if (v != 1 && v != 2 && v != 4 && v != 8) return false;
         t.unpack_tiles = (uint32_t)v;
         return true;
     }},
    {"FRIEDA_INTT_GENERIC", 0, 1, [](Tuning& t, long v) { t.intt_generic = v != 0; return true; }},
    {"FRIEDA_ERASURE_TREE_MIN_LOG", 6, 32, [](Tuning& t, long v) { t.erasure_tree_min_log = (uint32_t)v; return true; }},
    {"FRIEDA_TAIL_RUN_LOG", 4, 11, [](Tuning& t, long v) { t.tail_run_log = (uint32_t)v; return true; }},
    {"FRIEDA_HOST_DECOMMIT", 0, 1, [](Tuning& t, long v) { t.host_decommit = v != 0; return true; }},
    {"FRIEDA_GATHER_COPY", 0, 1, [](Tuning& t, long v) { t.gather_copy = v != 0; return true; }},
    {"FRIEDA_TREE_SKIP_LOG", 10, 40, [](Tuning& t, long v) { t.tree_skip_log = (uint32_t)v; return true; }},
    {"FRIEDA_TREE_SKIP_LONE_LOG", 10, 40, [](Tuning& t, long v) { t.tree_skip_lone_log = (uint32_t)v; return true; }},
    {"FRIEDA_TP_MIN_WGS", 0, 1 << 30, [](Tuning& t, long v) { t.tp_min_wgs = (uint32_t)v; return true; }},
    {"FRIEDA_GRIND_ITERS", 0, 64, [](Tuning& t, long v) { t.grind_iters = (uint32_t)v; return true; }},
    {"FRIEDA_BATCH_BUDGET_MB", 0, 262144, [](Tuning& t, long v) { t.batch_budget_mb = (uint32_t)v; return true; }},
    {"FRIEDA_BATCH_CALLS_PER_CTX", 1, 64, [](Tuning& t, long v) { t.batch_calls_per_ctx = (uint32_t)v; return true; }},
};
}  // namespace

bool tuning_set(Tuning& t, const char* name, long value) {
    if (!name) return false;
    for (const KnobDesc& k : KNOBS) {
        if (strcmp(k.name, name) == 0) {
            if (value < k.lo || value > k.hi) return false;
            return k.set(t, value);
        }
    }
    return false;
}

// Process-level FRIEDA_* variables that are read where they apply (not options of a context); DESIGN.md §10
static const char* const PROCESS_VARS[] = {"FRIEDA_HIP_LIB", "FRIEDA_RCCL_PATH", "FRIEDA_MULTI_FORCE_RCCL", "FRIEDA_MULTI_NO_PREFETCH",
                                           "FRIEDA_MULTI_NO_NUMA_PIN", "FRIEDA_BENCH_FORCE_DIST", "FRIEDA_BENCH_CPU_THREADS", "FRIEDA_TEST_"};
// A FRIEDA_* variable nobody reads is almost always a typo or a knob of an older round (FRIEDA_TEST_GRIND_FIRST_LOG became a test
// hook in round 5): say so once per process instead of silently doing nothing.
static void warn_unknown_env_once() {
    static std::once_flag once;
    std::call_once(once, [] {
        for (char** e = ::environ; e && *e; e++) {
            if (strncmp(*e, "FRIEDA_", 7) != 0) continue;
            const char* eq = strchr(*e, '=');
            const size_t len = eq ? (size_t)(eq - *e) : strlen(*e);
            bool known = false;
            for (const KnobDesc& k : KNOBS) known = known || (strlen(k.name) == len && strncmp(k.name, *e, len) == 0);
            for (const char* pv : PROCESS_VARS) known = known || strncmp(pv, *e, strlen(pv)) == 0;
            if (!known) fprintf(stderr, "libfrieda_hip: note: environment variable %.*s is not an option of this library (ignored)\n", (int)len, *e);
        }
    });
}

Tuning tuning_from_env() {
    Tuning t;
    warn_unknown_env_once();
    for (const KnobDesc& k : KNOBS) {
        const char* e = getenv(k.name);
        if (!e) continue;
        // a variable that is set but empty / not a number switches a boolean knob on, as `getenv(...) != nullptr` used to
        char* end = nullptr;
        long v = strtol(e, &end, 10);
        if (end == e) v = 1;
        (void)tuning_set(t, k.name, v);
    }
    return t;
}

const Tuning& tuning_defaults() {
    static const Tuning d;  // compiled-in defaults, immutable
    return d;
}
}  // namespace k

int Ctx::set_kernel_timing(bool enabled) {
    if (enabled && !timer) timer = new KernelTimerImpl();
    if (!enabled && timer) {
        (void)hipStreamSynchronize(stream);
        delete timer;
        timer = nullptr;
    }
    return FRIEDA_OK;
}

std::string Ctx::kernel_timing_report(bool reset) {
    std::string out = "{\"kernels\": [";
    if (timer) {
        (void)hipStreamSynchronize(stream);
        struct Agg {
            const char* name;
            size_t launches;
            double ms, bytes;
        };
        std::vector<Agg> agg;
        for (auto& sp : timer->spans) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, sp.start, sp.stop) != hipSuccess) continue;
            Agg* a = nullptr;
            for (auto& x : agg)
                if (strcmp(x.name, sp.name) == 0) a = &x;
            if (!a) {
                agg.push_back(Agg{sp.name, 0, 0.0, 0.0});
                a = &agg.back();
            }
            a->launches++;
            a->ms += ms;
            a->bytes += sp.alg_bytes;
        }
        bool first = true;
        for (auto& a : agg) {
            char buf[256];
            snprintf(buf, sizeof buf, "%s{\"name\": \"%s\", \"launches\": %zu, \"total_ms\": %.6f, \"alg_bytes\": %.1f}", first ? "" : ", ",
                     a.name, a.launches, a.ms, a.bytes);
            out += buf;
            first = false;
        }
        if (reset) {
            timer->spans.clear();
            timer->used = 0;
        }
    }
    out += "]}";
    return out;
}

// ---- Ctx ---------------------------------------------------------------------------------------
int Ctx::fail(int code, const std::string& what) {
    err = what;
    return code;
}
int Ctx::hip_fail(hipError_t e, const char* what) {
    err = std::string(what) + ": " + hipGetErrorString(e);
    return FRIEDA_ERR_HIP;
}

int Ctx::ensure_arena(size_t bytes) {
    if (tuning.test_arena_limit && bytes > tuning.test_arena_limit) {  // test hook: a device with less memory than this one
        err = "workspace of " + std::to_string(bytes) + " B refused by the test limit";
        return FRIEDA_ERR_NOMEM;
    }
    if (bytes <= arena_bytes) return FRIEDA_OK;
    if (arena) {
        FR_HIP(this, hipStreamSynchronize(stream));
        FR_HIP(this, hipFree(arena));
        arena = nullptr;
        arena_bytes = 0;
    }
    size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    hipError_t e = hipMalloc((void**)&arena, want);
    if (e != hipSuccess) {
        err = std::string("hipMalloc(workspace ") + std::to_string(want) + " B): " + hipGetErrorString(e);
        return FRIEDA_ERR_NOMEM;
    }
    arena_bytes = want;
    return FRIEDA_OK;
}

void Ctx::drop_twiddles() {
    for (auto& kv : twiddles) {
        if (kv.second.d_tw) (void)hipFree(kv.second.d_tw);
        if (kv.second.d_itw) (void)hipFree(kv.second.d_itw);
        if (kv.second.d_scratch) (void)hipFree(kv.second.d_scratch);
    }
    twiddles.clear();
}

int Ctx::get_twiddles(uint32_t n, TwiddleSet& out) {
    auto it = twiddles.find(n);
    if (it != twiddles.end() && cache_twiddles) {
        out = it->second;
        return FRIEDA_OK;
    }
    TwiddleSet ts;
    if (it != twiddles.end()) {
        ts = it->second;  // regenerate into the existing buffers (reference behaviour: recomputed per call)
    } else {
        size_t bytes = sizeof(uint32_t) << (n - 1);
        hipError_t e = hipMalloc((void**)&ts.d_tw, bytes);
        if (e == hipSuccess) e = hipMalloc((void**)&ts.d_itw, bytes);
        if (e == hipSuccess) e = hipMalloc((void**)&ts.d_scratch, 8192);
        if (e != hipSuccess) {  // nothing of a half-allocated set survives
            if (ts.d_tw) (void)hipFree(ts.d_tw);
            if (ts.d_itw) (void)hipFree(ts.d_itw);
            err = std::string("hipMalloc(twiddles, 2 x ") + std::to_string(bytes) + " B): " + hipGetErrorString(e);
            return FRIEDA_ERR_NOMEM;
        }
        twiddles[n] = ts;  // owned by the cache from here on, whatever happens below
    }
    // seeds: initial point of half_odds(n-1) and the step multiples the kernel combines
    Coset h = Coset::half_odds(n - 1);
    k::TwiddleSeeds seeds;
    memset(&seeds, 0, sizeof seeds);
    seeds.p0 = point_from_index(h.initial);
    if (n >= 3) {
        CPoint sp = point_from_index(h.step);
        for (uint32_t b = 0; b + 2 < n; b++) {
            seeds.step[b] = sp;
            sp = cp_double(sp);
        }
    }
    ts.ds.init_x = seeds.p0.x;
    ts.ds.init_y = seeds.p0.y;
    ts.ds.inv_init_x = m31_inv(seeds.p0.x);
    ts.ds.inv_init_y = m31_inv(seeds.p0.y);
    k::gen_twiddles(launch(), n, seeds, ts.d_tw, ts.d_itw, ts.d_scratch);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess) {
        // a cached entry must never describe tables that were not generated: drop it (and its buffers) before reporting
        (void)hipStreamSynchronize(stream);
        (void)hipFree(ts.d_tw);
        (void)hipFree(ts.d_itw);
        (void)hipFree(ts.d_scratch);
        twiddles.erase(n);
        return hip_fail(e, "gen_twiddles");
    }
    twiddles[n] = ts;
    out = ts;
    return FRIEDA_OK;
}

Ctx::~Ctx() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    drop_twiddles();
    delete timer;
    if (arena) (void)hipFree(arena);
    if (pinned) (void)hipHostFree(pinned);
    if (pinned_in) (void)hipHostFree(pinned_in);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
}

}  // namespace frieda
