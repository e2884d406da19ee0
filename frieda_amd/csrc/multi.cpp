// multi.cpp — a batch of independent blobs across the GPUs of one node, behind the C ABI (include/frieda_hip.h, "multi-GPU").
//
// What it replaces: a caller looping `api::commit` / `proof::commit_and_generate_proof` over blobs
// (/root/reference/src/lib.rs:31-38; the bench loops of benches/commit.rs:11-15, benches/proof.rs:30-44).  The blobs are
// independent, so the path shards at blob granularity: blob i -> device i mod n, one host thread and two
// contexts per device (two calls in flight: the latency chain of one runs under the wide kernels of the other; runs of
// equal-length blobs go through the batched kernels, as many per call as the batch policy's workspace budget allows, so the chain is also
// paid once per unit), no data-path
// collective.  The only exchange is the gather of the 32-byte commitment roots: one `ncclAllGather` per device on a
// single-process communicator (`ncclCommInitAll`) — RCCL over xGMI — after which every device holds every root (slot layout:
// rank-major, blob i at rank i mod n, slot i div n); the host reads device 0's copy.  With one device there is nothing to
// gather and RCCL is not touched (FRIEDA_MULTI_FORCE_RCCL=1 forces the one-rank collective: the hardware test of this file).
//
// RCCL is bound at frieda_multi_create by dlopen, not by a NEEDED entry: a process that already carries an RCCL (PyTorch
// bundles its own librccl.so.1) must share that instance, and single-GPU users of libfrieda_hip.so should not load a 570 MB
// library.  FRIEDA_RCCL_PATH overrides the library (the tests substitute a recording stub).
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <string.h>

#include <algorithm>
#include <cctype>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "host.h"

using namespace frieda;

namespace {

// ---- the slice of the RCCL API used here (signatures as in <rccl/rccl.h>) ----
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;  // ncclSuccess == 0
constexpr int kNcclUint8 = 1;  // ncclDataType_t::ncclUint8

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;

    bool load(std::string& err) {
        const char* override_path = getenv("FRIEDA_RCCL_PATH");
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (override_path && *override_path) {
            handle = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char* nm : names)  // an instance the process already carries wins
                if (!handle) handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            for (const char* nm : names)
                if (!handle) handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        }
        if (!handle) {
            const char* e = dlerror();
            err = std::string("RCCL not loadable (librccl.so.1): ") + (e ? e : "?");
            return false;
        }
        auto sym = [&](const char* nm) -> void* {
            void* p = dlsym(handle, nm);
            if (!p && err.empty()) err = std::string("RCCL lacks symbol ") + nm;
            return p;
        };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        return err.empty();
    }
};

}  // namespace

// ---- NUMA placement of the per-device host threads ----
// On a two-socket 8-GPU node each device hangs off one socket; its worker thread feeds it with hipMemcpyAsync from pageable
// memory (a staging copy by the runtime on the calling thread's CPU) and assembles its proofs.  Each worker is therefore pinned to
// the CPUs of its GPU's NUMA node: /sys/bus/pci/devices/<hipDeviceGetPCIBusId>/numa_node -> /sys/devices/system/node/node<k>/cpulist,
// intersected with the affinity the process was given.  Anything missing (no sysfs entry, node -1, empty intersection,
// FRIEDA_MULTI_NO_NUMA_PIN=1) leaves the thread where the scheduler puts it.
namespace frieda {
// "0-3,8,10-11\n" -> {0,1,2,3,8,10,11}; false on malformed text (nothing is pinned then)
bool parse_cpulist(const char* text, std::vector<int>& cpus) {
    cpus.clear();
    if (!text) return false;
    const char* p = text;
    while (*p == ' ' || *p == '\t') p++;
    if (*p == '\0' || *p == '\n') return true;  // an empty list is valid (a memory-only node)
    for (;;) {
        char* end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p || a < 0 || a > 65535) return false;
        long b = a;
        p = end;
        if (*p == '-') {
            p++;
            b = strtol(p, &end, 10);
            if (end == p || b < a || b > 65535) return false;
            p = end;
        }
        for (long c = a; c <= b; c++) cpus.push_back((int)c);
        if (*p == ',') {
            p++;
            continue;
        }
        while (*p == ' ' || *p == '\t' || *p == '\n') p++;
        return *p == '\0';
    }
}
}  // namespace frieda

namespace {
bool read_small_file(const std::string& path, std::string& out) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096];
    const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
    fclose(f);
    buf[n] = 0;
    out = buf;
    return true;
}
// the CPUs of the NUMA node of the PCI function `bus` ("0000:c1:00.0") under `sysfs` ("/sys"; a fake tree in the CPU tests), restricted to
// the process's affinity; empty: unknown / nothing to do
std::vector<int> cpus_near_bus_id(const std::string& sysfs, std::string bus) {
    std::vector<int> out;
    for (char& c : bus) c = (char)tolower(c);
    std::string node_s, list_s;
    if (!read_small_file(sysfs + "/bus/pci/devices/" + bus + "/numa_node", node_s)) return out;
    const long node = strtol(node_s.c_str(), nullptr, 10);
    if (node < 0) return out;  // -1: the platform reports no affinity
    if (!read_small_file(sysfs + "/devices/system/node/node" + std::to_string(node) + "/cpulist", list_s)) return out;
    std::vector<int> near;
    if (!frieda::parse_cpulist(list_s.c_str(), near)) return out;
    cpu_set_t have;
    CPU_ZERO(&have);
    if (sched_getaffinity(0, sizeof(have), &have) != 0) return out;
    for (int c : near)
        if (c < CPU_SETSIZE && CPU_ISSET(c, &have)) out.push_back(c);
    return out;
}
// the CPUs near HIP device `dev`
std::vector<int> cpus_near_device(int dev) {
    const char* off = getenv("FRIEDA_MULTI_NO_NUMA_PIN");
    if (off && *off == '1') return {};
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), dev) != hipSuccess) {
        (void)hipGetLastError();
        return {};
    }
    return cpus_near_bus_id("/sys", bus);
}
void pin_this_thread(const std::vector<int>& cpus) {
    if (cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus) CPU_SET(c, &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);  // best effort
}
}  // namespace

// joins whatever was started, on every exit path (a std::thread destroyed while joinable terminates the process)
struct JoinAll {
    std::vector<std::thread>& w;
    ~JoinAll() {
        for (auto& t : w)
            if (t.joinable()) t.join();
    }
};

// finishes (into scratch) whatever a context still has in flight, so that the handle stays usable after a failed call
static void drain_ctx(frieda_ctx* c) {
    try {
        const uint32_t cnt = job_count(&c->c);
        if (!cnt) return;
        std::vector<ProofData> scratch(cnt);
        std::vector<uint8_t> r(32 * (size_t)cnt);
        (void)prove_finish_batch(&c->c, r.data(), scratch);
    } catch (...) {
    }
}

inline size_t ring_stride(size_t len) { return (len + 255) & ~(size_t)255; }
constexpr uint32_t RING_SLOTS_C = 3;  // (= RING_SLOTS below)

// A device's share of the blobs (blob d, d + n, ...) cut into units = calls of the batched kernels.  A run of equal-length blobs is
// cut by the library's batch policy (host.h: workspace bytes in flight, two calls in flight per device), anything else is a single
// blob.  With host blobs travelling ahead of their kernels, the very first unit is a single blob when more follow: the chip starts
// after one upload instead of a whole unit's.
struct Unit {
    uint32_t slot, cnt;
};
//  The policy's budget is a figure for an otherwise empty MI355X.  `mem_cap` = the bytes this device can give the pass NOW
// (what hipMemGetInfo reports free + what the device's two contexts and its ring already hold, less a margin; 0 = unknown): two calls'
// workspaces and three ring slots must fit.  `shrink`: the per-call count halved that many times (the workers' retry after a
// FRIEDA_ERR_NOMEM that the estimate did not foresee).  `from_slot`: the device's first blob not yet done.
static void cut_device_units(const size_t* lens, uint32_t d, uint32_t n, uint32_t mine, bool batchable, bool prove, uint32_t log_blowup,
                             uint32_t log_last, const k::Tuning& tune, bool first_single, std::vector<Unit>& units, size_t& ring_need,
                             uint32_t from_slot = 0, uint64_t mem_cap = 0, uint32_t shrink = 0) {
    units.clear();
    ring_need = 0;
    std::vector<uint32_t> calls;
    for (uint32_t slot = from_slot; slot < mine;) {
        const uint32_t i0 = d + slot * n;
        uint32_t run = 1;
        while (batchable && slot + run < mine && lens[i0 + run * n] == lens[i0]) run++;
        const size_t ws = workspace_bytes_per_blob(lens[i0], log_blowup, log_last, prove, true);  // 0: the call itself reports the shape error
        uint32_t per = ws ? batch_per_call(tune, ws, run, 2) : 1u;
        if (ws && mem_cap) per = (uint32_t)std::min<uint64_t>(per, std::max<uint64_t>(1, mem_cap / (2 * (uint64_t)ws + RING_SLOTS_C * ring_stride(lens[i0]))));
        per = std::max<uint32_t>(1, per >> std::min<uint32_t>(shrink, 31));
        if (first_single && slot == from_slot && mine - from_slot > 1 && run > 1) {
            // Host blobs: a unit starts when its last blob has arrived, and the upload of unit u + 1 runs under the kernels of unit u.  A
            // pageable upload is about as slow per blob as the proof itself, so the units grow from one blob by at most half a unit at
            // a time (1, 2, 3, 5, 8, 12, ...) until they reach the policy's size: the chip starts after one upload and never waits for
            // a unit that is much longer to upload than its predecessor took to compute (32 blobs of 15.7 MB: 2.13 -> 1.9x ms per blob
            // pageable against units of 1, 16, 15).
            uint32_t sz = 1, left = run;
            while (left > 0 && sz < per) {
                const uint32_t c = std::min(sz, left);
                units.push_back(Unit{slot, c});
                ring_need = std::max(ring_need, ring_stride(lens[i0]) * c);
                slot += c;
                left -= c;
                sz = sz + (sz + 1) / 2;
            }
            run = left;
            if (run == 0) continue;
        }
        batch_cut(run, per, 2, calls);
        for (uint32_t c : calls) {
            units.push_back(Unit{slot, c});
            ring_need = std::max(ring_need, ring_stride(lens[i0]) * c);
            slot += c;
        }
    }
}

// Host blobs travel ahead of their kernels.  The two contexts of a device finish their units at about the same time (two launches
// of the same kernel share the chip evenly), so an upload enqueued on a context's own stream at `begin` would run with the chip
// idle (measured: 1.15 ms per unit of four 15.7 MB blobs, 0.3 ms per blob — profiles/r03_multi_trace_before.txt).  Instead
// each device has a copy stream and three staging slots in device memory: the blobs of unit u + 2 are uploaded while units u and
// u + 1 compute, an event orders the unit's first kernel behind its upload, and the kernels read the blobs where they landed.
constexpr uint32_t RING_SLOTS = RING_SLOTS_C;
struct UploadRing {
    hipStream_t stream = nullptr;
    hipEvent_t ready[RING_SLOTS] = {};
    uint8_t* slot[RING_SLOTS] = {};
    size_t cap = 0;  // bytes per slot
};

struct frieda_multi {
    std::vector<int> devices;
    std::vector<frieda_ctx*> ctx;        // 2 per device: [2 d], [2 d + 1]
    std::vector<UploadRing> ring;        // per device
    std::vector<std::vector<int>> near_cpus;  // per device: the CPUs of its NUMA node (empty: unknown, the worker is not pinned)
    bool prefetch = true;                // FRIEDA_MULTI_NO_PREFETCH=1: uploads on the contexts' own streams at begin (round 2's path; A/B knob)
    std::vector<hipStream_t> gstream;    // per device: the stream the gather runs on
    std::vector<uint8_t*> d_send, d_recv;
    size_t slot_cap = 0;                 // roots per device the gather buffers hold
    bool use_rccl = false;
    RcclApi rccl;
    std::vector<ncclComm_t> comms;
    std::string err;
    uint64_t gathers = 0;                // collectives issued so far (diagnostic)

    int fail(int code, const std::string& what) {
        err = what;
        return code;
    }
    int ensure_gather_buffers(size_t slots);
    // device d's worker thread only: slots of at least `bytes` each (grown between calls: nothing is in flight on the copy stream then)
    int ensure_ring(size_t d, size_t bytes, std::string& what);
    // enqueue the upload of `cnt` host blobs of `len` bytes (blob k at ptrs[k]) into slot `s` of device d; records ready[s]
    int upload(size_t d, uint32_t s, const uint8_t* const* ptrs, size_t len, uint32_t cnt, std::string& what);
    int gather_roots(const std::vector<std::vector<Hash32>>& local, uint32_t count, uint8_t* out_roots);
    // device d's worker thread only (the device is current).  Bytes a pass over this device may plan with: free now + held by the
    // device's two workspaces and its ring, less 10 %; 0 = the runtime would not say.
    uint64_t mem_cap_now(size_t d) const {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        const uint64_t held = (uint64_t)ctx[2 * d]->c.arena_bytes + ctx[2 * d + 1]->c.arena_bytes + (uint64_t)ring[d].cap * RING_SLOTS;
        return (fr + held) / 10 * 9;
    }
    // frees device d's two workspaces (nothing may be in flight on them): before a retry with smaller calls, and for
    // frieda_multi_release_workspace
    void drop_device_workspace(size_t d, bool ring_too) {
        for (frieda_ctx* c : {ctx[2 * d], ctx[2 * d + 1]}) {
            (void)hipStreamSynchronize(c->c.stream);
            if (c->c.arena) (void)hipFree(c->c.arena);
            c->c.arena = nullptr;
            c->c.arena_bytes = 0;
        }
        if (ring_too) {
            (void)hipStreamSynchronize(ring[d].stream);
            for (uint32_t k = 0; k < RING_SLOTS; k++) {
                if (ring[d].slot[k]) (void)hipFree(ring[d].slot[k]);
                ring[d].slot[k] = nullptr;
            }
            ring[d].cap = 0;
        }
    }
};

int frieda_multi::ensure_ring(size_t d, size_t bytes, std::string& what) {
    UploadRing& r = ring[d];
    if (bytes < 256) bytes = 256;
    if (r.cap >= bytes) return FRIEDA_OK;
    if (hipSetDevice(devices[d]) != hipSuccess || hipStreamSynchronize(r.stream) != hipSuccess) {
        what = "hipStreamSynchronize(upload stream)";
        return FRIEDA_ERR_HIP;
    }
    size_t cap = (size_t)1 << 20;
    while (cap < bytes) cap *= 2;
    for (uint32_t k = 0; k < RING_SLOTS; k++) {
        if (r.slot[k]) (void)hipFree(r.slot[k]);
        r.slot[k] = nullptr;
    }
    r.cap = 0;
    for (uint32_t k = 0; k < RING_SLOTS; k++)
        if (hipMalloc((void**)&r.slot[k], cap) != hipSuccess) {
            (void)hipGetLastError();
            what = "hipMalloc(upload ring, " + std::to_string(cap) + " B)";
            return FRIEDA_ERR_NOMEM;
        }
    r.cap = cap;
    return FRIEDA_OK;
}

int frieda_multi::upload(size_t d, uint32_t s, const uint8_t* const* ptrs, size_t len, uint32_t cnt, std::string& what) {
    UploadRing& r = ring[d];
    const size_t stride = ring_stride(len);
    if (stride * cnt > r.cap && len) {
        what = "internal: upload ring slot too small";
        return FRIEDA_ERR_INVARIANT;
    }
    for (uint32_t k = 0; k < cnt && len; k++)
        if (hipMemcpyAsync(r.slot[s] + k * stride, ptrs[k], len, hipMemcpyHostToDevice, r.stream) != hipSuccess) {
            what = "hipMemcpyAsync(blob upload)";
            return FRIEDA_ERR_HIP;
        }
    if (hipEventRecord(r.ready[s], r.stream) != hipSuccess) {
        what = "hipEventRecord(blob upload)";
        return FRIEDA_ERR_HIP;
    }
    return FRIEDA_OK;
}

int frieda_multi::ensure_gather_buffers(size_t slots) {
    if (slots <= slot_cap) return FRIEDA_OK;
    const size_t n = devices.size();
    size_t cap = 64;
    while (cap < slots) cap *= 2;
    for (size_t d = 0; d < n; d++) {
        if (hipSetDevice(devices[d]) != hipSuccess) return fail(FRIEDA_ERR_HIP, "hipSetDevice");
        if (d_send[d]) (void)hipFree(d_send[d]);
        if (d_recv[d]) (void)hipFree(d_recv[d]);
        d_send[d] = d_recv[d] = nullptr;
        if (hipMalloc((void**)&d_send[d], 32 * cap) != hipSuccess || hipMalloc((void**)&d_recv[d], 32 * cap * n) != hipSuccess) {
            slot_cap = 0;
            return fail(FRIEDA_ERR_NOMEM, "hipMalloc(root gather buffers)");
        }
    }
    slot_cap = cap;
    return FRIEDA_OK;
}

// local[d] = the roots device d produced, in its processing order (blob d, d + n, ...).  Every device ends up with every root;
// out_roots (host, count * 32 bytes, blob order) is read back from device 0.
int frieda_multi::gather_roots(const std::vector<std::vector<Hash32>>& local, uint32_t count, uint8_t* out_roots) {
    const size_t n = devices.size();
    const size_t per = (count + n - 1) / n;
    if (!use_rccl) {  // one device: nothing to exchange
        for (size_t s = 0; s < local[0].size(); s++) memcpy(out_roots + 32 * s, local[0][s].data(), 32);
        return FRIEDA_OK;
    }
    int rc = ensure_gather_buffers(per);
    if (rc) return rc;
    std::vector<uint8_t> stage(32 * per);
    for (size_t d = 0; d < n; d++) {
        if (hipSetDevice(devices[d]) != hipSuccess) return fail(FRIEDA_ERR_HIP, "hipSetDevice");
        memset(stage.data(), 0, stage.size());
        for (size_t s = 0; s < local[d].size(); s++) memcpy(stage.data() + 32 * s, local[d][s].data(), 32);
        if (hipMemcpy(d_send[d], stage.data(), stage.size(), hipMemcpyHostToDevice) != hipSuccess) return fail(FRIEDA_ERR_HIP, "hipMemcpy(roots H2D)");
    }
    ncclResult_t r = rccl.GroupStart();
    for (size_t d = 0; d < n && r == 0; d++) {
        if (hipSetDevice(devices[d]) != hipSuccess) return fail(FRIEDA_ERR_HIP, "hipSetDevice");
        r = rccl.AllGather(d_send[d], d_recv[d], 32 * per, kNcclUint8, comms[d], gstream[d]);
    }
    const ncclResult_t r2 = rccl.GroupEnd();
    if (r == 0) r = r2;
    if (r != 0) return fail(FRIEDA_ERR_HIP, std::string("ncclAllGather(roots): ") + rccl.GetErrorString(r));
    gathers++;
    for (size_t d = 0; d < n; d++) {
        if (hipSetDevice(devices[d]) != hipSuccess || hipStreamSynchronize(gstream[d]) != hipSuccess)
            return fail(FRIEDA_ERR_HIP, "hipStreamSynchronize(root gather)");
    }
    std::vector<uint8_t> all(32 * per * n);
    if (hipSetDevice(devices[0]) != hipSuccess || hipMemcpy(all.data(), d_recv[0], all.size(), hipMemcpyDeviceToHost) != hipSuccess)
        return fail(FRIEDA_ERR_HIP, "hipMemcpy(roots D2H)");
    for (uint32_t i = 0; i < count; i++) {
        const uint8_t* got = all.data() + 32 * ((size_t)(i % n) * per + i / n);
        if (memcmp(got, local[i % n][i / n].data(), 32) != 0) return fail(FRIEDA_ERR_INVARIANT, "internal: gathered root differs from the producer's copy");
        memcpy(out_roots + 32 * (size_t)i, got, 32);
    }
    return FRIEDA_OK;
}

extern "C" {

int frieda_multi_create(const int* devices, uint32_t n_devices, frieda_multi** out) {
    if (!out) return FRIEDA_ERR_ARG;
    *out = nullptr;
    if (!devices || n_devices == 0 || n_devices > 64) return FRIEDA_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return FRIEDA_ERR_HIP;
    for (uint32_t d = 0; d < n_devices; d++)
        if (devices[d] < 0 || devices[d] >= ndev) return FRIEDA_ERR_ARG;  // before any HIP call could record a sticky error
    frieda_multi* m = new (std::nothrow) frieda_multi();
    if (!m) return FRIEDA_ERR_NOMEM;
    try {
        m->devices.assign(devices, devices + n_devices);
        m->ctx.assign(2 * (size_t)n_devices, nullptr);
        m->gstream.assign(n_devices, nullptr);
        m->d_send.assign(n_devices, nullptr);
        m->d_recv.assign(n_devices, nullptr);
        m->ring.assign(n_devices, UploadRing{});
        m->near_cpus.resize(n_devices);
        for (uint32_t d = 0; d < n_devices; d++) m->near_cpus[d] = cpus_near_device(devices[d]);
        {
            const char* np = getenv("FRIEDA_MULTI_NO_PREFETCH");
            m->prefetch = !(np && *np == '1');
        }
        int rc = FRIEDA_OK;
        for (uint32_t d = 0; d < n_devices && rc == FRIEDA_OK; d++) {
            for (int k = 0; k < 2 && rc == FRIEDA_OK; k++) rc = frieda_ctx_create(devices[d], nullptr, &m->ctx[2 * d + k]);
            if (rc == FRIEDA_OK && (hipSetDevice(devices[d]) != hipSuccess || hipStreamCreateWithFlags(&m->gstream[d], hipStreamNonBlocking) != hipSuccess))
                rc = FRIEDA_ERR_HIP;
            if (rc == FRIEDA_OK && hipStreamCreateWithFlags(&m->ring[d].stream, hipStreamNonBlocking) != hipSuccess) rc = FRIEDA_ERR_HIP;
            for (uint32_t k = 0; k < RING_SLOTS && rc == FRIEDA_OK; k++)
                if (hipEventCreateWithFlags(&m->ring[d].ready[k], hipEventDisableTiming) != hipSuccess) rc = FRIEDA_ERR_HIP;
        }
        const char* force = getenv("FRIEDA_MULTI_FORCE_RCCL");
        m->use_rccl = n_devices > 1 || (force && *force == '1');
        if (rc == FRIEDA_OK && m->use_rccl) {
            std::string e;
            if (!m->rccl.load(e)) {
                rc = FRIEDA_ERR_HIP;
                fprintf(stderr, "frieda_multi_create: %s\n", e.c_str());
            } else {
                m->comms.assign(n_devices, nullptr);
                const ncclResult_t r = m->rccl.CommInitAll(m->comms.data(), (int)n_devices, m->devices.data());
                if (r != 0) {
                    fprintf(stderr, "frieda_multi_create: ncclCommInitAll: %s\n", m->rccl.GetErrorString(r));
                    m->comms.clear();
                    rc = FRIEDA_ERR_HIP;
                }
            }
        }
        if (rc != FRIEDA_OK) {
            frieda_multi_destroy(m);
            return rc;
        }
    } catch (...) {
        frieda_multi_destroy(m);
        return FRIEDA_ERR_NOMEM;
    }
    *out = m;
    return FRIEDA_OK;
}

int frieda_multi_destroy(frieda_multi* m) {
    if (!m) return FRIEDA_ERR_ARG;
    for (frieda_ctx*& c : m->ctx) {  // first: a context synchronises its stream, whose kernels may read the upload ring
        if (c) frieda_ctx_destroy(c);
        c = nullptr;
    }
    for (size_t d = 0; d < m->devices.size(); d++) {
        (void)hipSetDevice(m->devices[d]);
        // (each vector by its own size: creation may have failed between two of the assign() calls)
        if (d < m->comms.size() && m->comms[d]) (void)m->rccl.CommDestroy(m->comms[d]);
        if (d < m->gstream.size() && m->gstream[d]) {
            (void)hipStreamSynchronize(m->gstream[d]);
            (void)hipStreamDestroy(m->gstream[d]);
        }
        if (d < m->d_send.size() && m->d_send[d]) (void)hipFree(m->d_send[d]);
        if (d < m->d_recv.size() && m->d_recv[d]) (void)hipFree(m->d_recv[d]);
        if (d < m->ring.size()) {
            UploadRing& r = m->ring[d];
            if (r.stream) (void)hipStreamSynchronize(r.stream);
            for (uint32_t k = 0; k < RING_SLOTS; k++) {
                if (r.ready[k]) (void)hipEventDestroy(r.ready[k]);
                if (r.slot[k]) (void)hipFree(r.slot[k]);
            }
            if (r.stream) (void)hipStreamDestroy(r.stream);
        }
    }
    // the RCCL handle stays open: other users of the process (PyTorch) may share the instance
    delete m;
    return FRIEDA_OK;
}

uint32_t frieda_multi_device_count(const frieda_multi* m) { return m ? (uint32_t)m->devices.size() : 0; }
const char* frieda_multi_last_error(const frieda_multi* m) { return m ? m->err.c_str() : "null handle"; }
int frieda_multi_uses_rccl(const frieda_multi* m) { return m && m->use_rccl ? 1 : 0; }
uint64_t frieda_multi_gather_count(const frieda_multi* m) { return m ? m->gathers : 0; }
uint32_t frieda_multi_near_cpus(const frieda_multi* m, uint32_t device_slot, int* out_cpus, size_t cap) {
    if (!m || device_slot >= m->near_cpus.size()) return 0;
    const std::vector<int>& v = m->near_cpus[device_slot];
    for (size_t i = 0; i < v.size() && i < cap && out_cpus; i++) out_cpus[i] = v[i];
    return (uint32_t)v.size();
}
int frieda_test_near_cpus(const char* sysfs_root, const char* pci_bus_id, int* out_cpus, size_t cap, size_t* n) {
    if (!sysfs_root || !pci_bus_id || !n) return FRIEDA_ERR_ARG;
    *n = 0;
    try {
        const std::vector<int> v = cpus_near_bus_id(sysfs_root, pci_bus_id);
        *n = v.size();
        if (v.size() > cap) return FRIEDA_ERR_ARG;
        for (size_t i = 0; i < v.size() && out_cpus; i++) out_cpus[i] = v[i];
        return FRIEDA_OK;
    } catch (...) {
        return FRIEDA_ERR_NOMEM;
    }
}
int frieda_test_parse_cpulist(const char* text, int* out_cpus, size_t cap, size_t* n) {
    if (!n) return FRIEDA_ERR_ARG;
    *n = 0;
    try {
        std::vector<int> v;
        if (!parse_cpulist(text, v)) return FRIEDA_ERR_FORMAT;
        *n = v.size();
        if (v.size() > cap) return FRIEDA_ERR_ARG;
        for (size_t i = 0; i < v.size() && out_cpus; i++) out_cpus[i] = v[i];
        return FRIEDA_OK;
    } catch (...) {
        return FRIEDA_ERR_NOMEM;
    }
}
int frieda_multi_release_workspace(frieda_multi* m) {
    if (!m) return FRIEDA_ERR_ARG;
    for (size_t d = 0; d < m->devices.size(); d++) {
        if (hipSetDevice(m->devices[d]) != hipSuccess) return m->fail(FRIEDA_ERR_HIP, "hipSetDevice");
        for (frieda_ctx* c : {m->ctx[2 * d], m->ctx[2 * d + 1]})
            if (c->c.job || c->c.commit_pending) return m->fail(FRIEDA_ERR_ARG, "a call is in flight on this handle");
        m->drop_device_workspace(d, true);
    }
    return FRIEDA_OK;
}
frieda_ctx* frieda_multi_ctx(frieda_multi* m, uint32_t device_slot) {
    return m && device_slot < m->devices.size() ? m->ctx[2 * (size_t)device_slot] : nullptr;
}

int frieda_commit_many(frieda_multi* m, const uint8_t* const* blobs, const size_t* lens, uint32_t count, uint32_t log_blowup_factor,
                       uint8_t* out_roots) {
    if (!m) return FRIEDA_ERR_ARG;
    if (count == 0) return FRIEDA_OK;
    if (!blobs || !lens || !out_roots) return m->fail(FRIEDA_ERR_ARG, "null argument");
    for (uint32_t i = 0; i < count; i++)
        if (!blobs[i] && lens[i]) return m->fail(FRIEDA_ERR_ARG, "null blob");
    try {
        const size_t n = m->devices.size();
        std::vector<std::vector<Hash32>> local(n);
        std::vector<int> status(n, FRIEDA_OK);
        std::vector<std::string> what(n);
        std::atomic<bool> abort{false};
        std::vector<std::thread> workers;
        JoinAll join_guard{workers};
        for (size_t d = 0; d < n; d++) local[d].resize((count + n - 1 - d) / n);
        for (size_t d = 0; d < n; d++) {
            workers.emplace_back([&, d] {
                pin_this_thread(m->near_cpus[d]);
                frieda_ctx* cx[2] = {m->ctx[2 * d], m->ctx[2 * d + 1]};
                cx[1]->c.tuning.test_arena_limit = cx[0]->c.tuning.test_arena_limit;  // (test hook set through frieda_multi_ctx: the slot's both contexts)
                const bool pf = m->prefetch;
                struct DrainUploads {  // no upload may still be reading the caller's blobs when the worker returns
                    frieda_multi* m;
                    size_t d;
                    bool on;
                    ~DrainUploads() {
                        if (on) (void)hipStreamSynchronize(m->ring[d].stream);
                    }
                } drain_uploads{m, d, pf};
                try {
                    // This device's blobs in order, cut into units (cut_device_units: a run of blobs of one length = calls of the
                    // batched kernels, sized by the batch policy); two units in flight on the two contexts, the blobs of the unit after them being uploaded
                    // meanwhile on the copy stream.
                    const uint32_t mine = (uint32_t)local[d].size();
                    if (mine == 0) return;
                    if (hipSetDevice(m->devices[d]) != hipSuccess) {
                        status[d] = FRIEDA_ERR_HIP;
                        what[d] = "hipSetDevice";
                        abort.store(true);
                        return;
                    }
                    // One PASS over what is left of this device's blobs; a pass that ran out of device memory is repeated from its first
                    // unfinished blob with the calls halved (pass_st / pass_msg: the pass's own status, published when final).
                    uint32_t done = 0, shrink = 0;
                    for (;;) {
                    int pass_st = FRIEDA_OK;
                    std::string pass_msg;
                    uint32_t largest = 1;
                    auto pass = [&] {
                    std::vector<Unit> units;
                    size_t ring_need = 0;
                    cut_device_units(lens, (uint32_t)d, (uint32_t)n, mine, true, false, log_blowup_factor, 0, cx[0]->c.tuning, pf, units, ring_need, done,
                                     m->mem_cap_now(d), shrink);
                    for (const Unit& un : units) largest = std::max(largest, un.cnt);
                    auto bail_msg = [&](int rc, const std::string& msg) {
                        if (pass_st == FRIEDA_OK) {
                            pass_st = rc;
                            pass_msg = msg;
                        }
                    };
                    auto bail = [&](int rc, frieda_ctx* c) { bail_msg(rc, c->c.err); };
                    if (units.empty()) return;
                    std::string uw;
                    if (pf) {
                        const int rr = m->ensure_ring(d, ring_need, uw);
                        if (rr != FRIEDA_OK) return bail_msg(rr, uw);
                    }
                    auto upload = [&](size_t u) {
                        const uint32_t i0 = (uint32_t)d + units[u].slot * (uint32_t)n;
                        std::vector<const uint8_t*> ptrs(units[u].cnt);
                        for (uint32_t k = 0; k < units[u].cnt; k++) ptrs[k] = blobs[i0 + k * n];
                        return m->upload(d, (uint32_t)(u % RING_SLOTS), ptrs.data(), lens[i0], units[u].cnt, uw);
                    };
                    auto begin = [&](size_t u) {
                        const uint32_t i0 = (uint32_t)d + units[u].slot * (uint32_t)n;
                        Ctx* c = &cx[u & 1]->c;
                        if (pf) {
                            if (hipStreamWaitEvent(c->stream, m->ring[d].ready[u % RING_SLOTS], 0) != hipSuccess)
                                return c->fail(FRIEDA_ERR_HIP, "hipStreamWaitEvent(blob upload)");
                            return commit_batch_begin(c, m->ring[d].slot[u % RING_SLOTS], ring_stride(lens[i0]), lens[i0], units[u].cnt, true,
                                                      log_blowup_factor, nullptr);
                        }
                        std::vector<const uint8_t*> ptrs(units[u].cnt);
                        for (uint32_t k = 0; k < units[u].cnt; k++) ptrs[k] = blobs[i0 + k * n];
                        return commit_batch_begin(c, ptrs[0], lens[i0], lens[i0], units[u].cnt, false, log_blowup_factor, ptrs.data());
                    };
                    int rc = FRIEDA_OK;
                    for (size_t u = 0; pf && u < 2 && u < units.size() && rc == FRIEDA_OK; u++) rc = upload(u);
                    if (rc != FRIEDA_OK) return bail_msg(rc, uw);
                    rc = begin(0);
                    if (rc != FRIEDA_OK) return bail(rc, cx[0]);
                    for (size_t u = 0; u < units.size(); u++) {
                        bool next_begun = false;
                        if (u + 1 < units.size() && !abort.load()) {
                            rc = begin(u + 1);
                            if (rc != FRIEDA_OK) bail(rc, cx[(u + 1) & 1]);
                            next_begun = rc == FRIEDA_OK;
                        }
                        if (pf && u + 2 < units.size() && !abort.load()) {  // slot (u + 2) % 3 held unit u - 1, which has finished
                            rc = upload(u + 2);
                            if (rc != FRIEDA_OK) bail_msg(rc, uw);
                        }
                        const int rf = commit_batch_finish(&cx[u & 1]->c, local[d][units[u].slot].data());
                        if (rf != FRIEDA_OK)
                            bail(rf, cx[u & 1]);
                        else
                            done = units[u].slot + units[u].cnt;
                        if (pass_st != FRIEDA_OK || abort.load()) {
                            if (next_begun) {
                                std::vector<uint8_t> r(32 * (size_t)units[u + 1].cnt);
                                (void)commit_batch_finish(&cx[(u + 1) & 1]->c, r.data());
                            }
                            return;
                        }
                    }
                    };  // pass
                    pass();
                    if (pass_st == FRIEDA_ERR_NOMEM && largest > 1 && !abort.load() && shrink < 16) {
                        m->drop_device_workspace(d, pf);  // (synchronises the contexts' streams and the copy stream first)
                        shrink++;
                        continue;
                    }
                    if (pass_st != FRIEDA_OK) {
                        status[d] = pass_st;
                        what[d] = pass_msg;
                        abort.store(true);
                    }
                    return;
                    }  // passes
                } catch (...) {  // nothing may unwind out of a worker thread
                    what[d] = "host allocation failed";
                    status[d] = FRIEDA_ERR_NOMEM;
                    abort.store(true);
                    for (frieda_ctx* c : cx)
                        if (c->c.commit_pending) {
                            (void)hipStreamSynchronize(c->c.stream);
                            c->c.commit_pending = 0;
                        }
                }
            });
        }
        for (auto& w : workers) w.join();
        for (size_t d = 0; d < n; d++)
            if (status[d] != FRIEDA_OK) return m->fail(status[d], "device " + std::to_string(m->devices[d]) + ": " + what[d]);
        return m->gather_roots(local, count, out_roots);
    } catch (const std::bad_alloc&) {
        return m->fail(FRIEDA_ERR_NOMEM, "host allocation failed");
    } catch (const std::exception& e) {
        return m->fail(FRIEDA_ERR_INVARIANT, e.what());
    }
}

int frieda_prove_many(frieda_multi* m, const uint8_t* const* blobs, const size_t* lens, uint32_t count, const uint64_t* seeds,
                      frieda_pcs_config cfg, uint8_t* out_commitments, frieda_proof** out_proofs) {
    if (!m) return FRIEDA_ERR_ARG;
    if (count == 0) return FRIEDA_OK;
    if (!blobs || !lens || !out_commitments || !out_proofs) return m->fail(FRIEDA_ERR_ARG, "null argument");
    for (uint32_t i = 0; i < count; i++) {
        out_proofs[i] = nullptr;
        if (!blobs[i] && lens[i]) return m->fail(FRIEDA_ERR_ARG, "null blob");
    }
    try {
        const size_t n = m->devices.size();
        std::vector<std::vector<Hash32>> local(n);
        std::vector<int> status(n, FRIEDA_OK);
        std::vector<std::string> what(n);
        std::atomic<bool> abort{false};
        std::vector<std::thread> workers;
        JoinAll join_guard{workers};
        for (size_t d = 0; d < n; d++) local[d].resize((count + n - 1 - d) / n);
        for (size_t d = 0; d < n; d++) {
            workers.emplace_back([&, d] {
              pin_this_thread(m->near_cpus[d]);
              const bool pf = m->prefetch;
              struct DrainUploads {  // no upload may still be reading the caller's blobs when the worker returns
                  frieda_multi* m;
                  size_t d;
                  bool on;
                  ~DrainUploads() {
                      if (on) (void)hipStreamSynchronize(m->ring[d].stream);
                  }
              } drain_uploads{m, d, pf};
              try {
                // This device's blobs in order, cut into units (cut_device_units: a run of blobs of one length goes through the
                // batched kernels, as many per call as the batch policy's workspace budget allows — the Fiat-Shamir chain is paid
                // once per unit —, anything else is a single proof).  Two units in
                // flight: begin(u + 1) is enqueued on the other context before finish(u) waits; the blobs of unit u + 2 are
                // uploaded on the copy stream meanwhile.
                frieda_ctx* cx[2] = {m->ctx[2 * d], m->ctx[2 * d + 1]};
                cx[1]->c.tuning.test_arena_limit = cx[0]->c.tuning.test_arena_limit;  // (test hook set through frieda_multi_ctx: the slot's both contexts)
                const uint32_t mine = (uint32_t)local[d].size();
                if (mine == 0) return;
                // batches need the device transcript (prover.cpp): small last layers, and neither context switched to the host channel
                // through frieda_multi_ctx (then equal-length runs go as single proofs instead of failing)
                const bool batchable = cfg.log_last_layer_degree_bound <= 11 && cfg.log_blowup_factor <= 11 &&
                                       cfg.log_last_layer_degree_bound + cfg.log_blowup_factor <= 11 && !cx[0]->c.host_channel &&
                                       !cx[1]->c.host_channel;
                if (hipSetDevice(m->devices[d]) != hipSuccess) {
                    status[d] = FRIEDA_ERR_HIP;
                    what[d] = "hipSetDevice";
                    abort.store(true);
                    return;
                }
                // passes over what is left of this device's blobs: see frieda_commit_many (a pass that ran out of device memory is
                // repeated from its first unfinished blob with the calls halved)
                uint32_t done = 0, shrink = 0;
                for (;;) {
                int pass_st = FRIEDA_OK;
                std::string pass_msg;
                uint32_t largest = 1;
                auto pass = [&] {
                std::vector<Unit> units;
                size_t ring_need = 0;
                cut_device_units(lens, (uint32_t)d, (uint32_t)n, mine, batchable, true, cfg.log_blowup_factor, cfg.log_last_layer_degree_bound,
                                 cx[0]->c.tuning, pf, units, ring_need, done, m->mem_cap_now(d), shrink);
                for (const Unit& un : units) largest = std::max(largest, un.cnt);
                auto bail_msg = [&](int rc, const std::string& msg) {
                    if (pass_st == FRIEDA_OK) {
                        pass_st = rc;
                        pass_msg = msg;
                    }
                };
                auto bail = [&](int rc, frieda_ctx* c) { bail_msg(rc, c->c.err); };
                if (units.empty()) return;
                std::string uw;
                if (pf) {
                    const int rr = m->ensure_ring(d, ring_need, uw);
                    if (rr != FRIEDA_OK) return bail_msg(rr, uw);
                }
                auto upload = [&](size_t u) {
                    const uint32_t i0 = (uint32_t)d + units[u].slot * (uint32_t)n;
                    std::vector<const uint8_t*> ptrs(units[u].cnt);
                    for (uint32_t k = 0; k < units[u].cnt; k++) ptrs[k] = blobs[i0 + k * n];
                    return m->upload(d, (uint32_t)(u % RING_SLOTS), ptrs.data(), lens[i0], units[u].cnt, uw);
                };
                auto begin = [&](size_t u) {
                    const Unit& un = units[u];
                    const uint32_t i0 = (uint32_t)d + un.slot * (uint32_t)n;
                    Ctx* c = &cx[u & 1]->c;
                    std::vector<const uint8_t*> ptrs_v(un.cnt);
                    std::vector<uint64_t> sd_v(un.cnt);
                    const uint8_t** ptrs = ptrs_v.data();
                    uint64_t* sd = sd_v.data();
                    for (uint32_t k = 0; k < un.cnt; k++) {
                        ptrs[k] = blobs[i0 + k * n];
                        sd[k] = seeds ? seeds[i0 + k * n] : 0;
                    }
                    if (pf) {
                        if (hipStreamWaitEvent(c->stream, m->ring[d].ready[u % RING_SLOTS], 0) != hipSuccess)
                            return c->fail(FRIEDA_ERR_HIP, "hipStreamWaitEvent(blob upload)");
                        const uint8_t* dev = m->ring[d].slot[u % RING_SLOTS];
                        if (un.cnt == 1) return prove_begin(c, dev, lens[i0], true, seeds ? &seeds[i0] : nullptr, cfg);
                        return prove_begin_batch(c, dev, ring_stride(lens[i0]), lens[i0], un.cnt, true, seeds ? sd : nullptr, cfg);
                    }
                    if (un.cnt == 1) return prove_begin(c, blobs[i0], lens[i0], false, seeds ? &seeds[i0] : nullptr, cfg);
                    return prove_begin_batch_ptrs(c, ptrs, lens[i0], un.cnt, seeds ? sd : nullptr, cfg);
                };
                auto finish = [&](size_t u) {  // -> status; fills local roots and out_proofs of the unit
                    const Unit& un = units[u];
                    frieda_ctx* c = cx[u & 1];
                    std::vector<frieda_proof*> objs(un.cnt, nullptr);
                    std::vector<ProofData> outs(un.cnt);
                    for (uint32_t k = 0; k < un.cnt; k++) {
                        objs[k] = c->pool->get();
                        objs[k]->home = c->pool;
                        outs[k] = std::move(objs[k]->p);
                    }
                    const int rf = prove_finish_batch(&c->c, local[d][un.slot].data(), outs);
                    for (uint32_t k = 0; k < un.cnt; k++) {
                        if (k < outs.size()) objs[k]->p = std::move(outs[k]);
                        if (rf != FRIEDA_OK)
                            c->pool->put(objs[k]);
                        else
                            out_proofs[(uint32_t)d + (un.slot + k) * (uint32_t)n] = objs[k];
                    }
                    return rf;
                };
                int rc = FRIEDA_OK;
                for (size_t u = 0; pf && u < 2 && u < units.size() && rc == FRIEDA_OK; u++) rc = upload(u);
                if (rc != FRIEDA_OK) return bail_msg(rc, uw);
                rc = begin(0);
                if (rc != FRIEDA_OK) return bail(rc, cx[0]);
                for (size_t u = 0; u < units.size(); u++) {
                    bool next_begun = false;
                    if (u + 1 < units.size() && !abort.load()) {
                        rc = begin(u + 1);
                        if (rc != FRIEDA_OK) bail(rc, cx[(u + 1) & 1]);
                        next_begun = rc == FRIEDA_OK;
                    }
                    if (pf && u + 2 < units.size() && !abort.load()) {  // slot (u + 2) % 3 held unit u - 1, which has finished
                        rc = upload(u + 2);
                        if (rc != FRIEDA_OK) bail_msg(rc, uw);
                    }
                    const int rf = finish(u);
                    if (rf != FRIEDA_OK)
                        bail(rf, cx[u & 1]);
                    else
                        done = units[u].slot + units[u].cnt;
                    if (pass_st != FRIEDA_OK || abort.load()) {
                        if (next_begun) drain_ctx(cx[(u + 1) & 1]);  // the unit already enqueued: the context must stay reusable
                        return;
                    }
                }
                };  // pass
                pass();
                if (pass_st == FRIEDA_ERR_NOMEM && largest > 1 && !abort.load() && shrink < 16) {
                    m->drop_device_workspace(d, pf);  // (synchronises the contexts' streams and the copy stream first)
                    shrink++;
                    continue;
                }
                if (pass_st != FRIEDA_OK) {
                    status[d] = pass_st;
                    what[d] = pass_msg;
                    abort.store(true);
                }
                return;
                }  // passes
              } catch (...) {  // nothing may unwind out of a worker thread
                  status[d] = FRIEDA_ERR_NOMEM;
                  what[d] = "host allocation failed";
                  abort.store(true);
                  drain_ctx(m->ctx[2 * d]);
                  drain_ctx(m->ctx[2 * d + 1]);
              }
            });
        }
        for (auto& w : workers) w.join();
        int rc = FRIEDA_OK;
        for (size_t d = 0; d < n && rc == FRIEDA_OK; d++)
            if (status[d] != FRIEDA_OK) rc = m->fail(status[d], "device " + std::to_string(m->devices[d]) + ": " + what[d]);
        if (rc == FRIEDA_OK && abort.load()) rc = m->fail(FRIEDA_ERR_INVARIANT, "internal: aborted without a status");
        if (rc == FRIEDA_OK) rc = m->gather_roots(local, count, out_commitments);
        if (rc != FRIEDA_OK)
            for (uint32_t i = 0; i < count; i++) {
                if (out_proofs[i]) frieda_proof_free(out_proofs[i]);
                out_proofs[i] = nullptr;
            }
        return rc;
    } catch (const std::bad_alloc&) {
        // (the workers have been joined by JoinAll; whatever they published is handed back to nobody)
        for (uint32_t i = 0; i < count; i++) {
            if (out_proofs[i]) frieda_proof_free(out_proofs[i]);
            out_proofs[i] = nullptr;
        }
        return m->fail(FRIEDA_ERR_NOMEM, "host allocation failed");
    } catch (const std::exception& e) {
        for (uint32_t i = 0; i < count; i++) {
            if (out_proofs[i]) frieda_proof_free(out_proofs[i]);
            out_proofs[i] = nullptr;
        }
        return m->fail(FRIEDA_ERR_INVARIANT, e.what());
    }
}

}  // extern "C"
