// handoff_probe.hip — what it costs on MI355X to hand a tree level from many workgroups to one (and an alpha back to many):
// a dependent kernel boundary against the in-kernel forms a persistent "narrow phase" kernel would use.  Measurement aid for
// DESIGN.md §5 (latency chain), not part of the product.
//
// One ROUND imitates one FRI layer of the narrow regime: W producer workgroups each publish one 32-byte hash, one consumer reads
// all W of them, and every workgroup then learns a 16-byte value (the alpha) the consumer derived from them.
//   A  launches      producers kernel -> consumer kernel -> (next round's producers read the alpha): two kernel boundaries
//   B  last arriver  ONE launch per round: producers store sc1 + drain + agent atomic add; the workgroup whose add came last
//                    consumes in place (one boundary per round)
//   C  persistent    ONE launch for all rounds: last arriver consumes, publishes the alpha + an epoch flag (sc1); everybody
//                    else polls the flag (relaxed sc1 loads + s_sleep), bounded spin
// No real hashing is done (the compute chain is the same in all three forms); each phase only moves the bytes, so the
// per-round time is the synchronisation + data hand-off price.  Results are checked (every round's value depends on all
// hashes of that round).
// Build: hipcc -O3 --offload-arch=gfx950 tools/handoff_probe.hip -o tools/handoff_probe.bin ; run: tools/handoff_probe.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            std::exit(1);                                                         \
        }                                                                         \
    } while (0)

typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

struct State {
    unsigned int counter[64];  // one arrival counter per round (zeroed by the host before every run)
    unsigned int epoch;        // persistent form: last completed round + 1
    unsigned int timeout;
    unsigned int alpha[4];
    unsigned int pad[58];
};

__device__ __forceinline__ unsigned int mixw(unsigned int a, unsigned int b) { return (a ^ (b * 0x9E3779B1u)) * 0x85EBCA6Bu + 1u; }

// the "hash" a producer publishes in round r, derived from the alpha of round r - 1 (so rounds are truly dependent)
__device__ __forceinline__ unsigned int produced(unsigned int wg, unsigned int round, unsigned int alpha0, unsigned int word) {
    return mixw(mixw(wg, round), alpha0 + word);
}

// ---- A: two kernels per round ----
__global__ void a_produce(unsigned int* hashes, const State* st, unsigned int round) {
    const unsigned int a0 = st->alpha[0];
    if (threadIdx.x < 8) hashes[8 * blockIdx.x + threadIdx.x] = produced(blockIdx.x, round, a0, threadIdx.x);
}
__global__ void a_consume(const unsigned int* hashes, State* st, unsigned int W) {
    __shared__ unsigned int s[256];
    unsigned int acc = 0;
    for (unsigned int i = threadIdx.x; i < 8 * W; i += 256) acc += hashes[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x < 4) st->alpha[threadIdx.x] = s[0] + threadIdx.x;
}

// ---- B / C shared pieces ----
__device__ __forceinline__ void publish_hash_sc1(unsigned int* hashes, unsigned int wg, unsigned int round, unsigned int a0) {
    if (threadIdx.x < 8)
        __hip_atomic_store((gu32*)(hashes + 8 * wg + threadIdx.x), produced(wg, round, a0, threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the arrival is signalled
    __syncthreads();
}
__device__ __forceinline__ unsigned int consume_sc1(const unsigned int* hashes, unsigned int W, unsigned int* s) {
    unsigned int acc = 0;
    for (unsigned int i = threadIdx.x; i < 8 * W; i += 256) acc += __hip_atomic_load((gu32*)(hashes + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
        __syncthreads();
    }
    return s[0];
}

// ---- B: one launch per round, the last arriver consumes ----
__global__ void b_round(unsigned int* hashes, State* st, unsigned int W, unsigned int round) {
    __shared__ unsigned int s[256];
    __shared__ unsigned int s_last;
    const unsigned int a0 = st->alpha[0];  // written by the previous launch: a kernel boundary orders it
    publish_hash_sc1(hashes, blockIdx.x, round, a0);
    if (threadIdx.x == 0) s_last = atomicAdd(&st->counter[round], 1u) == W - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    const unsigned int sum = consume_sc1(hashes, W, s);
    if (threadIdx.x < 4) st->alpha[threadIdx.x] = sum + threadIdx.x;
}

// ---- C: one launch for all rounds ----
__global__ void c_persistent(unsigned int* hashes, State* st, unsigned int W, unsigned int rounds) {
    __shared__ unsigned int s[256];
    __shared__ unsigned int s_last, s_a0;
    unsigned int a0 = 0;
    for (unsigned int r = 0; r < rounds; r++) {
        publish_hash_sc1(hashes + (size_t)(r & 1) * 8 * W, blockIdx.x, r, a0);
        if (threadIdx.x == 0) s_last = atomicAdd(&st->counter[r], 1u) == W - 1 ? 1u : 0u;
        __syncthreads();
        if (s_last) {
            const unsigned int sum = consume_sc1(hashes + (size_t)(r & 1) * 8 * W, W, s);
            if (threadIdx.x < 4) __hip_atomic_store((gu32*)&st->alpha[threadIdx.x], sum + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store((gu32*)&st->epoch, r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x == 0) {
            unsigned int spins = 0;
            while (__hip_atomic_load((gu32*)&st->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r + 1) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) {  // bounded: give up loudly instead of hanging the chip
                    st->timeout = 1;
                    break;
                }
            }
            s_a0 = __hip_atomic_load((gu32*)&st->alpha[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        a0 = s_a0;
        if (st->timeout) return;
    }
}

// keeps the stream busy while the host enqueues a whole run, so that the rounds are timed on the GPU side (as in the product,
// where every launch of a proof is enqueued long before it runs)
__global__ void delay_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

static unsigned int host_expect(unsigned int W, unsigned int rounds) {
    auto mixw_h = [](unsigned int a, unsigned int b) { return (a ^ (b * 0x9E3779B1u)) * 0x85EBCA6Bu + 1u; };
    unsigned int a0 = 0;
    for (unsigned int r = 0; r < rounds; r++) {
        unsigned int sum = 0;
        for (unsigned int wg = 0; wg < W; wg++)
            for (unsigned int w = 0; w < 8; w++) sum += mixw_h(mixw_h(wg, r), a0 + w);
        a0 = sum;
    }
    return a0;
}

int main() {
    const unsigned int rounds = 32;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0));
    CK(hipEventCreate(&ev1));
    unsigned int* hashes;
    State* st;
    CK(hipMalloc(&hashes, 2 * 8 * 512 * sizeof(unsigned int)));
    CK(hipMalloc(&st, sizeof(State)));
    std::printf("one round = W producers -> 1 consumer -> alpha back to all; %u dependent rounds per run; us per round (median of 20 runs)\n", rounds);
    std::printf("%6s %14s %16s %14s\n", "W", "A launches(2/r)", "B last-arriver", "C persistent");
    for (unsigned int W : {1u, 8u, 32u, 128u, 256u, 512u}) {
        double med[3];
        for (int form = 0; form < 3; form++) {
            std::vector<double> ts;
            for (int rep = 0; rep < 22; rep++) {
                CK(hipMemsetAsync(st, 0, sizeof(State), s));
                CK(hipStreamSynchronize(s));
                delay_kernel<<<1, 64, 0, s>>>(100000ull);  // ~1 ms at the 100 MHz real-time counter
                CK(hipEventRecord(ev0, s));
                if (form == 0) {
                    for (unsigned int r = 0; r < rounds; r++) {
                        a_produce<<<W, 256, 0, s>>>(hashes, st, r);
                        a_consume<<<1, 256, 0, s>>>(hashes, st, W);
                    }
                } else if (form == 1) {
                    for (unsigned int r = 0; r < rounds; r++) b_round<<<W, 256, 0, s>>>(hashes, st, W, r);
                } else {
                    c_persistent<<<W, 256, 0, s>>>(hashes, st, W, rounds);
                }
                CK(hipEventRecord(ev1, s));
                CK(hipStreamSynchronize(s));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, ev0, ev1));
                const double us = 1e3 * ms;
                State h;
                CK(hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost));
                if (h.timeout || h.alpha[0] != host_expect(W, rounds)) {
                    std::printf("form %d W %u: WRONG RESULT (timeout %u)\n", form, W, h.timeout);
                    return 1;
                }
                if (rep >= 2) ts.push_back(us / rounds);
            }
            std::sort(ts.begin(), ts.end());
            med[form] = ts[ts.size() / 2];
        }
        std::printf("%6u %14.2f %16.2f %14.2f\n", W, med[0], med[1], med[2]);
    }
    return 0;
}
