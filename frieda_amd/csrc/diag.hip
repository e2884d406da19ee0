// diag.hip — measurement aid behind frieda_ctx_blake2s_ceiling: the pure-compute rate of the Merkle compression on THIS device, now.
//
// The commit / prove path is bound by the integer VALU rate of Blake2s.  bench.py measures that ceiling in the same
// process, on the same device, right after the timed run, instead of quoting a number taken on another box (devices differ by a few
// per cent), and reads the clock inside the kernel with it: the chip holds ~2.39 GHz under this load (MI355X), so the ceiling is the
// instruction stream (~3250 / ~3480 SIMD cycles per leaf / node wave-compression in the throughput form of blake2s.h, which is what this
// kernel runs — the same code as the tree kernels), not a lowered clock: every lane chains compressions on register-resident data
// (no memory traffic), 8 workgroups per CU, once with the 4-word message of a leaf (12 zero words constant-folded) and once with
// the full 16-word message of an inner node.  (tools/blake2s_rate.hip is the stand-alone sweep over occupancies.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "blake2s.h"
#include "kernels.h"

namespace frieda {
namespace k {

namespace {
// Both counters in one volatile asm (s_memtime: shader clock; s_memrealtime: 100 MHz wall clock), waited for at once.  Volatile asm
// statements keep their order among themselves; the empty ones around them tie the stamp to the data flow of the loop it brackets
// (the loop's input depends on the first stamp, the second stamp follows an asm that consumes the loop's output), so the compiler
// can move neither across the loop.
__device__ __forceinline__ void stamp_pair(unsigned long long& c, unsigned long long& r) {
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r)::"memory");
}

// stamps: one {shader-clock, 100 MHz wall-clock} pair per workgroup at the start and at the end of the chain (wave 0), written to
// a buffer of their own that nothing else reads (MI355X_MICROARCH.md, DVFS give-back (6): in-kernel clock = d(s_memtime) /
// d(s_memrealtime) x 100 MHz).  This is a diagnostic kernel; no product kernel carries stamps.
template <int LEAF>
__global__ __launch_bounds__(256) void b2_chain_kernel(uint32_t* out, int iters, unsigned long long* stamps) {
    uint32_t m[16], h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 16; i++) m[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    unsigned long long c0 = 0, r0 = 0, c1 = 0, r1 = 0;
    if (stamps) {
        stamp_pair(c0, r0);
        asm volatile("" : "+v"(m[0]) : "s"(c0));  // the chain starts after the stamp
    }
    for (int it = 0; it < iters; it++) {
        if (LEAF) {
            b2_merkle_leaf<FRIEDA_B2_IDLE_LEAF>(m[0], m[1], m[2], m[3], h);  // (the leaf shape's own setting of the throughput form, as treedev::leaf_hash)
            for (int i = 0; i < 4; i++) m[i] = h[i] ^ h[4 + i];
        } else {
            b2_merkle_block(m, h);
            for (int i = 0; i < 8; i++) {
                m[i] ^= h[i];
                m[8 + i] += h[i];
            }
        }
    }
    if (stamps) {
        asm volatile("" ::"v"(h[0]), "v"(h[7]));  // the stamp follows the chain
        stamp_pair(c1, r1);
        if (threadIdx.x == 0) {
            stamps[4 * blockIdx.x + 0] = c0;
            stamps[4 * blockIdx.x + 1] = r0;
            stamps[4 * blockIdx.x + 2] = c1;
            stamps[4 * blockIdx.x + 3] = r1;
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += h[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;  // keeps the chain alive
}
}  // namespace

// d_scratch: >= blake2s_ceiling_scratch_bytes().  Returns compressions per second (0 on a HIP error), leaf- and node-shaped; with
// `clock` != null also, per shape, {in-kernel clock in GHz, SIMD cycles per wave-compression} (medians over the workgroups' stamps:
// cycles = d(s_memtime) / iterations / co-resident waves per SIMD).
size_t blake2s_ceiling_scratch_bytes() { return (size_t)256 * 8 * 256 * 4 + (size_t)256 * 8 * 4 * sizeof(unsigned long long); }

int blake2s_ceiling(hipStream_t s, uint32_t* d_scratch, double* leaf_per_s, double* node_per_s, double* clock) {
    const int blocks = 256 * 8, iters = 96, waves_per_simd = 8;
    unsigned long long* d_stamps = reinterpret_cast<unsigned long long*>(d_scratch + (size_t)blocks * 256);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
    double rate[2] = {0, 0};
    int rc = 0;
    std::vector<unsigned long long> st((size_t)blocks * 4);
    for (int leaf = 0; leaf < 2 && rc == 0; leaf++) {
        // warm-up launches (the chip settles on its clock under this load within a few milliseconds), the last one is timed
        // (with `clock`: ~0.25 s of that load first — a burst of a few tens of milliseconds reads a boost clock, 2.7 GHz on one MI355X,
        // that the chip does not sustain; after a quarter of a second it sits at the 2.39-2.40 GHz of profiles/r03_clock_probe_mi355x.txt)
        const int reps = clock ? 200 : 3;
        for (int rep = 0; rep < reps; rep++) {
            if (rep == reps - 1) (void)hipEventRecord(e0, s);
            unsigned long long* stp = (clock && rep == reps - 1) ? d_stamps : nullptr;
            if (leaf)
                b2_chain_kernel<1><<<blocks, 256, 0, s>>>(d_scratch, iters, stp);
            else
                b2_chain_kernel<0><<<blocks, 256, 0, s>>>(d_scratch, iters, stp);
        }
        (void)hipEventRecord(e1, s);
        float ms = 0.f;
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) rc = 1;
        else rate[leaf] = (double)blocks * 256.0 * iters / (ms * 1e-3);
        if (rc == 0 && clock) {
            if (hipMemcpy(st.data(), d_stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) {
                rc = 1;
                break;
            }
            std::vector<double> ghz, cyc;
            for (int b = 0; b < blocks; b++) {
                const double dc = (double)(st[4 * b + 2] - st[4 * b + 0]), dr = (double)(st[4 * b + 3] - st[4 * b + 1]);
                if (dr > 0) ghz.push_back(dc / dr * 0.1), cyc.push_back(dc);
            }
            if (ghz.empty()) {
                rc = 1;
                break;
            }
            std::sort(ghz.begin(), ghz.end());
            const double g = ghz[ghz.size() / 2];
            // cycles per wave-compression per SIMD from the wall rate and the clock: 1024 SIMDs x 64 lanes
            clock[2 * (1 - leaf) + 0] = g;
            clock[2 * (1 - leaf) + 1] = rate[leaf] > 0 ? 1024.0 * 64.0 * g * 1e9 / rate[leaf] : 0.0;
            (void)waves_per_simd;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (leaf_per_s) *leaf_per_s = rate[1];
    if (node_per_s) *node_per_s = rate[0];
    return rc;
}

}  // namespace k
}  // namespace frieda
