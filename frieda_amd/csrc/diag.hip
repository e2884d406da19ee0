// diag.hip — measurement aid behind frieda_ctx_blake2s_ceiling: the pure-compute rate of the Merkle compression on THIS device, now.
//
// The commit / prove path is bound by the integer VALU rate of Blake2s, and that rate is not a constant of the
// architecture: the chip lowers its clock under this all-lanes integer load, and devices differ by up to ~12 %
// (MI355X_MICROARCH.md, DVFS give-back).  bench.py therefore measures the ceiling in the same process, on the same device, right
// after the timed run, instead of quoting a number taken on another box: every lane chains compressions on register-resident data
// (no memory traffic), 8 workgroups per CU, once with the 4-word message of a leaf (12 zero words constant-folded) and once with
// the full 16-word message of an inner node.  (tools/blake2s_rate.hip is the stand-alone sweep over occupancies.)
#include <hip/hip_runtime.h>

#include "blake2s.h"
#include "kernels.h"

namespace frieda {
namespace k {

namespace {
template <int LEAF>
__global__ __launch_bounds__(256) void b2_chain_kernel(uint32_t* out, int iters) {
    uint32_t m[16], h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 16; i++) m[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    for (int it = 0; it < iters; it++) {
        if (LEAF) {
            const uint32_t mm[16] = {m[0], m[1], m[2], m[3], 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            b2_merkle_block(mm, h);
            for (int i = 0; i < 4; i++) m[i] = h[i] ^ h[4 + i];
        } else {
            b2_merkle_block(m, h);
            for (int i = 0; i < 8; i++) {
                m[i] ^= h[i];
                m[8 + i] += h[i];
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += h[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;  // keeps the chain alive
}
}  // namespace

// d_scratch: >= 256 * 8 * 256 * 4 bytes.  Returns compressions per second (0 on a HIP error), leaf- and node-shaped.
int blake2s_ceiling(hipStream_t s, uint32_t* d_scratch, double* leaf_per_s, double* node_per_s) {
    const int blocks = 256 * 8, iters = 96;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
    double rate[2] = {0, 0};
    int rc = 0;
    for (int leaf = 0; leaf < 2 && rc == 0; leaf++) {
        for (int rep = 0; rep < 3; rep++) {  // two warm-up launches (clock), the third is timed
            if (rep == 2) (void)hipEventRecord(e0, s);
            if (leaf)
                b2_chain_kernel<1><<<blocks, 256, 0, s>>>(d_scratch, iters);
            else
                b2_chain_kernel<0><<<blocks, 256, 0, s>>>(d_scratch, iters);
        }
        (void)hipEventRecord(e1, s);
        float ms = 0.f;
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) rc = 1;
        else rate[leaf] = (double)blocks * 256.0 * iters / (ms * 1e-3);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (leaf_per_s) *leaf_per_s = rate[1];
    if (node_per_s) *node_per_s = rate[0];
    return rc;
}

}  // namespace k
}  // namespace frieda
