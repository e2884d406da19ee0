// blake2s_variants.hip — does the instruction mix of the Blake2s G function change the chip-wide compression rate on MI355X?
// The slot count is the same for every variant below (18 full-rate slots per G), so any difference is the clock the chip
// sustains under that mix (the hash kernels run power-limited at ~1.75 GHz).  Measurement aid.
// Build: hipcc -O3 --offload-arch=gfx950 tools/blake2s_variants.hip -o tools/blake2s_variants.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__constant__ uint8_t SIG[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
constexpr uint8_t SIGC[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};

template <int V>
__device__ __forceinline__ uint32_t rot(uint32_t x, int r) {
    if (V == 2 && (r == 16 || r == 8)) {  // byte-granular rotations through v_perm_b32
        return r == 16 ? __builtin_amdgcn_perm(x, x, 0x01000302u) : __builtin_amdgcn_perm(x, x, 0x00030201u);
    }
    if (V == 3 && (r == 16 || r == 8)) return __builtin_amdgcn_alignbyte(x, x, r / 8);
    return __builtin_rotateright32(x, r);
}
template <int V>
__device__ __forceinline__ uint32_t add3(uint32_t a, uint32_t b, uint32_t c) {
    if (V == 1) {  // two VOP2 adds instead of one v_add3_u32
        uint32_t t;
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b));
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(c));
        return t;
    }
    return a + b + c;
}
#define G(a, b, c, d, x, y)             \
    do {                                \
        a = add3<V>(a, b, (x));         \
        d = rot<V>(d ^ a, 16);          \
        c = c + d;                      \
        b = rot<V>(b ^ c, 12);          \
        a = add3<V>(a, b, (y));         \
        d = rot<V>(d ^ a, 8);           \
        c = c + d;                      \
        b = rot<V>(b ^ c, 7);           \
    } while (0)

template <int V, int R>
__device__ __forceinline__ void round_(uint32_t (&v)[16], const uint32_t (&m)[16]) {
    G(v[0], v[4], v[8], v[12], m[SIGC[R][0]], m[SIGC[R][1]]);
    G(v[1], v[5], v[9], v[13], m[SIGC[R][2]], m[SIGC[R][3]]);
    G(v[2], v[6], v[10], v[14], m[SIGC[R][4]], m[SIGC[R][5]]);
    G(v[3], v[7], v[11], v[15], m[SIGC[R][6]], m[SIGC[R][7]]);
    G(v[0], v[5], v[10], v[15], m[SIGC[R][8]], m[SIGC[R][9]]);
    G(v[1], v[6], v[11], v[12], m[SIGC[R][10]], m[SIGC[R][11]]);
    G(v[2], v[7], v[8], v[13], m[SIGC[R][12]], m[SIGC[R][13]]);
    G(v[3], v[4], v[9], v[14], m[SIGC[R][14]], m[SIGC[R][15]]);
}
template <int V>
__device__ __forceinline__ void compress(const uint32_t (&m)[16], uint32_t (&h)[8]) {
    uint32_t v[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
    round_<V, 0>(v, m); round_<V, 1>(v, m); round_<V, 2>(v, m); round_<V, 3>(v, m); round_<V, 4>(v, m);
    round_<V, 5>(v, m); round_<V, 6>(v, m); round_<V, 7>(v, m); round_<V, 8>(v, m); round_<V, 9>(v, m);
    for (int i = 0; i < 8; i++) h[i] = v[i] ^ v[i + 8];
}

template <int V>
__global__ __launch_bounds__(256) void chain_kernel(uint32_t* out, int iters) {
    uint32_t m[16], h[8] = {};
    for (int i = 0; i < 16; i++) m[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    for (int it = 0; it < iters; it++) {
        compress<V>(m, h);
        for (int i = 0; i < 8; i++) {
            m[i] ^= h[i];
            m[8 + i] += h[i];
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += h[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
void run(const char* name) {
    uint32_t* d;
    const int blocks = 256 * 8;
    (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    const int iters = 1000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    chain_kernel<V><<<blocks, 256>>>(d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    chain_kernel<V><<<blocks, 256>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %7.2f G compressions/s\n", name, (double)blocks * 256 * iters / (ms * 1e-3) / 1e9);
    (void)hipFree(d);
}

int main() {
    for (int rep = 0; rep < 2; rep++) {
        run<0>("v_add3_u32 + v_alignbit_b32 (product)");
        run<1>("two v_add_u32 + v_alignbit_b32");
        run<2>("v_add3_u32 + v_perm_b32 for rot 16 / 8");
        run<3>("v_add3_u32 + v_alignbyte_b32 for rot 16 / 8");
    }
    return 0;
}
