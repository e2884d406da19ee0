// rccl_stub.cpp — TEST DOUBLE of the RCCL entry points frieda_multi binds (tests only; FRIEDA_RCCL_PATH points libfrieda_hip.so at it).
// It lets the one-GPU test box run the N > 1 root gather of frieda_amd/csrc/multi.cpp: the communicator accepts the same device
// several times (real RCCL refuses duplicates) and the grouped all-gather is done with device-to-device copies at ncclGroupEnd.
// It checks what multi.cpp must get right — one AllGather per rank inside one group, equal send counts, ncclUint8, distinct
// receive buffers — and fails the call otherwise.  It says nothing about RCCL itself: that is covered by the real one-rank
// collective (FRIEDA_MULTI_FORCE_RCCL=1) here and by the driver's multi-GPU bench.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

struct ncclComm {
    int rank, nranks, device;
    struct Group* group;
};
struct Group {
    std::vector<ncclComm*> members;
};
struct Pending {
    const void* send;
    void* recv;
    size_t count;
    int dtype;
    ncclComm* comm;
    hipStream_t stream;
};
static thread_local std::vector<Pending> g_pending;
static thread_local int g_depth = 0;
static int g_allgathers = 0, g_groups = 0;

extern "C" {
int frieda_rccl_stub_allgathers() { return g_allgathers; }
int frieda_rccl_stub_groups() { return g_groups; }

int ncclCommInitAll(ncclComm** comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1) return 4;  // ncclInvalidArgument
    Group* g = new Group();
    for (int r = 0; r < ndev; r++) {
        comms[r] = new ncclComm{r, ndev, devlist ? devlist[r] : r, g};
        g->members.push_back(comms[r]);
    }
    return 0;
}
int ncclCommDestroy(ncclComm* c) {
    delete c;  // the Group leaks: test double
    return 0;
}
int ncclGroupStart() {
    g_depth++;
    return 0;
}
static int flush() {
    if (g_pending.empty()) return 0;
    const int n = g_pending[0].comm->nranks;
    if ((int)g_pending.size() != n) {
        fprintf(stderr, "rccl_stub: %zu all-gathers in a group of %d ranks\n", g_pending.size(), n);
        return 5;  // ncclInvalidUsage
    }
    std::vector<const Pending*> by_rank(n, nullptr);
    for (const Pending& p : g_pending) {
        if (p.dtype != 1 || p.count != g_pending[0].count || p.comm->group != g_pending[0].comm->group || by_rank[p.comm->rank]) return 5;
        by_rank[p.comm->rank] = &p;
    }
    for (int r = 0; r < n; r++)
        for (int q = 0; q < n; q++)
            if (r != q && by_rank[r]->recv == by_rank[q]->recv) return 5;
    for (int r = 0; r < n; r++) {
        if (hipSetDevice(by_rank[r]->comm->device) != hipSuccess) return 1;
        for (int s = 0; s < n; s++)
            if (hipMemcpyAsync((char*)by_rank[r]->recv + (size_t)s * by_rank[r]->count, by_rank[s]->send, by_rank[r]->count, hipMemcpyDeviceToDevice,
                               by_rank[r]->stream) != hipSuccess)
                return 1;
    }
    g_allgathers += n;
    g_groups++;
    g_pending.clear();
    return 0;
}
int ncclGroupEnd() {
    if (--g_depth > 0) return 0;
    return flush();
}
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, ncclComm* comm, hipStream_t stream) {
    if (!send || !recv || !comm) return 4;
    g_pending.push_back(Pending{send, recv, count, dtype, comm, stream});
    if (g_depth == 0) return flush();
    return 0;
}
const char* ncclGetErrorString(int r) { return r == 0 ? "no error" : (r == 5 ? "stub: invalid usage" : "stub: error"); }
}
