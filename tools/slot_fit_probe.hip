// slot_fit_probe.hip — why the two contexts of the measured loop fall into lockstep.  Stream W keeps the chip full of long-lived workgroups
// with the footprint of a wide product kernel (256 threads, ~64 VGPRs, 12.8 KB LDS, 8 per CU, ~100 us each, thousands queued); stream C
// runs a chain of dependent one-workgroup kernels with (a) a footprint that fits the slot ONE retiring wide workgroup frees (256 threads,
// < 64 VGPRs, little LDS) and (b) the footprint of top_kernel (512 threads, ~112 VGPRs, 68 KB LDS), which needs the resources of several
// wide workgroups of the same CU at once.  Reported: time of the chain alone and under the wide stream.
// Build: hipcc -O3 --offload-arch=gfx950 tools/slot_fit_probe.hip -o tools/slot_fit_probe.bin
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>

__global__ __launch_bounds__(256) void wide_kernel(uint32_t* out, int iters) {
    __shared__ uint32_t lds[3200];  // 12.8 KB
    uint32_t r[48];
    for (int i = 0; i < 48; i++) r[i] = threadIdx.x * 2654435761u + i + blockIdx.x;
    lds[threadIdx.x] = r[0];
    __syncthreads();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 48; i++) r[i] = (r[i] ^ r[(i + 1) % 48]) + 0x9E3779B9u;  // keeps ~50 VGPRs live
    }
    uint32_t s = lds[(threadIdx.x + 1) & 255];
    for (int i = 0; i < 48; i++) s ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int THREADS, int NREG, int LDS_WORDS>
__global__ __launch_bounds__(THREADS) void narrow_kernel(uint32_t* inout, int iters) {
    __shared__ uint32_t lds[LDS_WORDS];
    uint32_t r[NREG];
    const uint32_t seed = inout[0];
    for (int i = 0; i < NREG; i++) r[i] = seed + threadIdx.x * 40503u + i;
    lds[threadIdx.x] = r[0];
    __syncthreads();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NREG; i++) r[i] = (r[i] ^ r[(i + 1) % NREG]) + 0x85EBCA6Bu;
    }
    uint32_t s = lds[(threadIdx.x + 7) % THREADS];
    for (int i = 0; i < NREG; i++) s ^= r[i];
    if (threadIdx.x == 0) inout[0] = s;  // the next kernel of the chain depends on it
}

template <typename F>
static double chain_ms(F launch, hipStream_t s, int n) {
    (void)hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) launch();
    (void)hipStreamSynchronize(s);
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char**) {
    hipStream_t sw, sc;
    // argv[1] = "prio": the chain's stream at the highest priority, the wide stream at the lowest
    int lo_p = 0, hi_p = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);
    const bool prio = argc > 1;
    if (prio) {
        printf("stream priorities: wide %d, chain %d\n", lo_p, hi_p);
        (void)hipStreamCreateWithPriority(&sw, hipStreamNonBlocking, lo_p);
        (void)hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, hi_p);
    } else {
        (void)hipStreamCreateWithFlags(&sw, hipStreamNonBlocking);
        (void)hipStreamCreateWithFlags(&sc, hipStreamNonBlocking);
    }
    uint32_t *d_out, *d_chain;
    const int wide_blocks = 256 * 8 * 16;  // 16 rounds of the chip's 2048 slots
    (void)hipMalloc(&d_out, (size_t)wide_blocks * 256 * 4);
    (void)hipMalloc(&d_chain, 64);
    (void)hipMemset(d_chain, 0, 64);
    const int wide_iters = 1500, n_chain = 100, narrow_iters = 20;
    auto small = [&] { narrow_kernel<256, 24, 1024><<<1, 256, 0, sc>>>(d_chain, narrow_iters); };
    auto big = [&] { narrow_kernel<512, 96, 17000><<<1, 512, 0, sc>>>(d_chain, narrow_iters); };
    // warm-up and the wide kernel's own time
    wide_kernel<<<wide_blocks, 256, 0, sw>>>(d_out, wide_iters);
    (void)hipStreamSynchronize(sw);
    const auto t0 = std::chrono::steady_clock::now();
    wide_kernel<<<wide_blocks, 256, 0, sw>>>(d_out, wide_iters);
    (void)hipStreamSynchronize(sw);
    const double wide_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    small();
    big();
    (void)hipStreamSynchronize(sc);
    const double small_alone = chain_ms(small, sc, n_chain), big_alone = chain_ms(big, sc, n_chain);
    printf("wide kernel: %d workgroups of 256 threads, %.2f ms alone (%.0f us per round of 2048)\n", wide_blocks, wide_ms, 1e3 * wide_ms / 16);
    for (int rep = 0; rep < 2; rep++) {
        for (int which = 0; which < 2; which++) {
            for (int k = 0; k < 4; k++) wide_kernel<<<wide_blocks, 256, 0, sw>>>(d_out, wide_iters);
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            const double under = which == 0 ? chain_ms(small, sc, n_chain) : chain_ms(big, sc, n_chain);
            (void)hipStreamSynchronize(sw);
            printf("chain of %d one-workgroup kernels, %s: %.2f ms alone, %.2f ms while the wide stream runs\n", n_chain,
                   which == 0 ? "256 threads / ~30 VGPRs / 4 KB LDS (fits one freed slot)" : "512 threads / ~100 VGPRs / 68 KB LDS (top_kernel's footprint)",
                   which == 0 ? small_alone : big_alone, under);
        }
    }
    return 0;
}
