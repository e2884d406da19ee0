// ntt.hip — Reed–Solomon encode: radix-2 circle FFT over M31 (gfx950).
//
// Replaces `SecureCirclePoly::evaluate_with_twiddles` -> 4 x `CpuBackend::evaluate`
// (/root/reference/src/commit.rs:16, src/proof.rs:48-49; stwo backend/cpu/circle.rs::evaluate,
// core/fft.rs::butterfly).  Input: 2^L coefficients per column in natural order (zero-extended to 2^n by
// the reference); output: 2^n evaluations per column in bit-reversed domain order.
//
// Structure.  Layer i pairs indices that differ in bit i; the twiddle of a pair depends only on the index
// bits above i.  Layers run i = n-1 .. 1 (line layers, twiddle T_{i-1}[idx >> (i+1)]) then i = 0 (circle
// layer, twiddle Y[idx >> 1]).  Because the coefficient vector is zero above 2^L, the top n-L layers are
// butterflies against zero: they only replicate the coefficient block 2^(n-L) times.  They are never
// executed — the first real pass reads coefficient `idx mod 2^L` instead.
//
// Passes.  A pass executes a run of layers i_hi..i_lo on a 4096-word tile per column held in LDS: the tile is
// the 2^t (t = i_hi-i_lo+1) values of index bits [i_lo, i_hi] times 2^log_w consecutive values of the low bits
// (so global accesses are 64-byte runs), all other index bits fixed per workgroup.  The last pass has i_lo = 0
// and reads/writes fully contiguous 16 KiB tiles with 16-byte accesses.
//
// Stages.  Inside a pass the layers are executed four at a time: a thread reads the 16 tile elements that differ
// in the stage's four index bits into registers, runs 4 x 8 butterflies on them, and writes them back — three LDS
// round trips for 12 layers instead of twelve.  The 15 twiddles of a stage depend only on the other index bits, so
// they are fetched once and reused for all columns of the workgroup (the 4 coordinate columns share the domain).
// LDS addresses are padded by idx >> 4, which makes every stage's 32-lane access conflict free.
//
// Cost.  One butterfly = v_mad_u64_u32 + Mersenne fold + two modular add/sub = 17 full-rate VALU slots (multiplies,
// v_alignbit and v_min are half rate on gfx950); 4 columns x L x N/2 butterflies make the encode VALU-bound at about
// twice its HBM time.
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cstdlib>

#include "clock_stamps.h"
#include "kernels.h"
#include "tree_dev.h"

namespace frieda {
namespace k {

namespace {
using namespace treedev;

constexpr int NTT_THREADS = 256;
constexpr uint32_t TILE_LOG = 12;
#ifndef FRIEDA_NTT_PAD32  // the default: one pad word per 16 elements, stage groups in thread order (rounds 1 - 6); -DFRIEDA_NTT_PAD32: the conflict-free layout below
constexpr uint32_t TILE_WORDS = (1u << TILE_LOG) + (1u << (TILE_LOG - 4));
#else
constexpr uint32_t TILE_WORDS = (1u << TILE_LOG) + (1u << (TILE_LOG - 5));  // padded (pad below)
#endif
constexpr uint32_t MAX_COLS_PER_WG = 4;
constexpr uint32_t MID_LOG_W = 4;  // 64-byte contiguous runs in the strided passes

// Tile element e sits at LDS word e + (e >> 4).  Round 6 found that this layout is NOT conflict-free (below) and built one that is
// (-DFRIEDA_NTT_PAD32: e + (e >> 5) and stage_group_base's dealing of the stage on bits 4 .. 7): SQ_LDS_BANK_CONFLICT 0.33 - 0.40 -> 0.000 of
// the LDS-array cycles, the strided pass -1.6 us, the contiguous pass unchanged, the fused last pass + tree launch of commitments 1 - 3 %
// SLOWER — the LDS array is not what these kernels wait for.  The default therefore stays; the build flag keeps the A/B
// (profiles/r06_lds_conflicts.txt, tools/lds_conflicts.sh).  The PAD32 layout:  ds_read_b32 / ds_write_b32 / the two halves of a
// ds_*2_b32 are served per 32-lane half with 32 banks (word mod 32).  With one pad word per 16 elements the stage on tile bits 8 .. 11 (a
// half-wave touches 32 CONSECUTIVE elements: words e .. e + 15, e + 17 .. e + 32) and the 16-byte tile fill hit one bank twice in every
// access: SQ_LDS_BANK_CONFLICT was a third of SQ_LDS_IDX_ACTIVE in ntt_tile12<3,0> and 40 % in ntt_tile12_rep (profiles/r06_lds_conflicts.txt).
// With one pad word per 32 elements those two patterns and the stage on bits 0 .. 3 (16 g + r -> 16 g + (g >> 1) + r) are conflict-free,
// and the stage on bits 4 .. 7 is once its groups are dealt to the threads by stage_group_base below.
#ifndef FRIEDA_NTT_PAD32
__device__ __forceinline__ uint32_t pad(uint32_t e) { return e + (e >> 4); }
#else
__device__ __forceinline__ uint32_t pad(uint32_t e) { return e + (e >> 5); }
#endif
// The group of 16 elements thread g (< 256) handles in the radix-16 stage on tile bits lo .. lo + 3 (lo = 8, 4, 0): its base element.
// lo = 4: the group with tile bits 0 .. 3 = g & 15 and bits 8 .. 11 = h, where h is g >> 4 with its two low bits swapped — a half-wave
// (g >> 4 = 2 m, 2 m + 1) then reads 16 elements at h and 16 at h + 2, 512 elements = 528 words = 16 banks apart, instead of h and h + 1
// (264 words = 8 banks apart: 8 of 16 banks twice).  A wave still owns the elements [1024 w, 1024 w + 1024) in this stage and the next.
__device__ __forceinline__ uint32_t stage_group_base(uint32_t g, uint32_t lo) {
#ifdef FRIEDA_NTT_PAD32
    if (lo == 4) {
        const uint32_t t = g >> 4;
        const uint32_t h = (t & ~3u) | ((t & 1u) << 1) | ((t >> 1) & 1u);
        return (h << 8) | (g & 15u);
    }
#endif
    return ((g >> lo) << (lo + 4)) | (g & ((1u << lo) - 1));
}

// ---- the four layers of a radix-16 stage on 16 register-resident values, eight butterflies at a time ----
// Same idea as the hash kernels' throughput form (blake2s.h): the eight independent butterflies of a layer advance together, one
// arithmetic step at a time, so that the VALU stream is made of runs of one rate class (v_mad_u64_u32 / v_min: slow; v_and / v_add /
// v_sub / shifts: fast), pinned by data flow, and the wave runs its slow runs at a raised priority (s_setprio in the pin statements):
// the SIMD then overlaps one wave's slow instruction with another wave's fast one (blake2s.h).  FRIEDA_NTT_IDLE = 0xPAB: P = priority
// of the slow runs (0 = no switching), A / B = idle issue states (s_nop) after a slow / after a fast run — the first form of round 5,
// 0x33, superseded by the priorities: 0x200; 0 = the plain form (butterfly by butterfly, the scheduler's order).  Everything of a
// transform kernel that is NOT a butterfly (loads, LDS traffic, stores) runs at FRIEDA_NTT_OUTER_PRIO (ntt_enter).
// The twiddles of these stages are kept DOUBLED in their registers (2 tw mod 2^32 = 2 tw): one instruction fewer per butterfly.
#ifndef FRIEDA_NTT_IDLE
#define FRIEDA_NTT_IDLE 0x200
#endif
#ifndef FRIEDA_NTT_GROUP
#define FRIEDA_NTT_GROUP 8  // butterflies advanced together (8 = a whole layer; 4 halves the temporaries)
#endif
#ifndef FRIEDA_NTT_GROUP_FUSED
#define FRIEDA_NTT_GROUP_FUSED 4  // in the kernels that keep all four columns in registers (120+ VGPRs)
#endif
// SET: the wave priority the statement sets (-1: leaves it alone) — bits 8 - 9 of FRIEDA_NTT_IDLE = the priority of the slow runs,
// the fast runs at 0 (see blake2s.h b2_pin)
// what a transform kernel runs at outside its butterflies (loads, LDS traffic, stores)
#ifndef FRIEDA_NTT_OUTER_PRIO
#define FRIEDA_NTT_OUTER_PRIO 3
#endif
__device__ __forceinline__ void ntt_enter() {
    if constexpr (FRIEDA_NTT_OUTER_PRIO != 0 && ((FRIEDA_NTT_IDLE >> 8) & 3) != 0) __builtin_amdgcn_s_setprio(FRIEDA_NTT_OUTER_PRIO);
}
template <int N, int G, int SET = -1>
__device__ __forceinline__ void ntt_pin(uint32_t (&a)[G]) {
    static_assert(N >= 0 && N <= 5 && (G == 4 || G == 8) && SET >= -1 && SET <= 3, "idle states 0 .. 5, groups of 4 or 8, priorities 0 .. 3");
#define FR_PIN_OPS4 "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
#define FR_PIN_OPS8 "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
#define FR_PIN_N(PRE, ...)                                               \
    do {                                                                 \
        if constexpr (N <= 1) asm volatile(PRE : __VA_ARGS__);           \
        if constexpr (N == 2) asm volatile(PRE "s_nop 0" : __VA_ARGS__); \
        if constexpr (N == 3) asm volatile(PRE "s_nop 1" : __VA_ARGS__); \
        if constexpr (N == 4) asm volatile(PRE "s_nop 2" : __VA_ARGS__); \
        if constexpr (N == 5) asm volatile(PRE "s_nop 3" : __VA_ARGS__); \
    } while (0)
#define FR_PIN_SET(...)                                                   \
    do {                                                                  \
        if constexpr (SET < 0 && N >= 1) FR_PIN_N("", __VA_ARGS__);       \
        if constexpr (SET == 0) FR_PIN_N("s_setprio 0\n\t", __VA_ARGS__); \
        if constexpr (SET == 1) FR_PIN_N("s_setprio 1\n\t", __VA_ARGS__); \
        if constexpr (SET == 2) FR_PIN_N("s_setprio 2\n\t", __VA_ARGS__); \
        if constexpr (SET == 3) FR_PIN_N("s_setprio 3\n\t", __VA_ARGS__); \
    } while (0)
    if constexpr (G == 8)
        FR_PIN_SET(FR_PIN_OPS8);
    else
        FR_PIN_SET(FR_PIN_OPS4);
#undef FR_PIN_SET
#undef FR_PIN_N
#undef FR_PIN_OPS4
#undef FR_PIN_OPS8
}
// G butterflies (k0 .. k0 + G - 1 of the layer's eight) of layer Q (0 = the stage's top bit): x[r], x[r | bit] with twiddle
// tw[(1 << Q) - 1 + (r >> (bit + 1))]
template <int IDLE, int Q, int G, int K0, typename TW>
__device__ __forceinline__ void radix16_group(uint32_t (&x)[16], const TW& tw) {
    constexpr int bit = 3 - Q, NA = (IDLE >> 4) & 15, NB = IDLE & 15, PRIO = (IDLE >> 8) & 3;
    constexpr int LO = PRIO ? 0 : -1, HI = PRIO ? PRIO : -1;  // what the statements after a slow / after a fast run set
    // the eight (low, high) index pairs of a layer
    constexpr int LOW[4][8] = {{0, 1, 2, 3, 4, 5, 6, 7}, {0, 1, 2, 3, 8, 9, 10, 11}, {0, 1, 4, 5, 8, 9, 12, 13}, {0, 2, 4, 6, 8, 10, 12, 14}};
    uint32_t hi[G], lo[G], d[G];
#pragma unroll
    for (int k = 0; k < G; k++) {
        const int r = LOW[Q][K0 + k];
        // the twiddle arrives DOUBLED (2 tw < 2^32): x * 2 tw = (x tw >> 31) * 2^32 + 2 (x tw mod 2^31), so the high dword of the
        // product IS the high part of the Mersenne fold and the low part is one shift away (no 64-bit shift, no mask)
        const uint64_t p = (uint64_t)x[r | (1 << bit)] * tw[(1 << Q) - 1 + (r >> (bit + 1))];
        hi[k] = (uint32_t)(p >> 32);
        lo[k] = (uint32_t)p;
    }
    ntt_pin<NA, G, LO>(hi);  // (lo follows from the same multiply: pinning one of the two results orders both)
#pragma unroll
    for (int k = 0; k < G; k++) {
        lo[k] >>= 1;
        lo[k] += hi[k];  // <= (P - 1) + P
        hi[k] = lo[k] - P31;
    }
    ntt_pin<NB, G, HI>(hi);
#pragma unroll
    for (int k = 0; k < G; k++) lo[k] = umin32(lo[k], hi[k]);  // t = x_high * tw, canonical
    ntt_pin<NA, G, LO>(lo);
#pragma unroll
    for (int k = 0; k < G; k++) {
        const int r = LOW[Q][K0 + k];
        const uint32_t w = x[r];
        d[k] = w - lo[k];
        x[r] = w + lo[k];
        hi[k] = d[k] + P31;
        lo[k] = x[r] - P31;
    }
    ntt_pin<NB, G, HI>(lo);
#pragma unroll
    for (int k = 0; k < G; k++) {
        const int r = LOW[Q][K0 + k];
        x[r] = umin32(x[r], lo[k]);
        x[r | (1 << bit)] = umin32(d[k], hi[k]);
    }
}
template <int IDLE, int Q, int G, typename TW>
__device__ __forceinline__ void radix16_layer(uint32_t (&x)[16], const TW& tw) {
    radix16_group<IDLE, Q, G, 0>(x, tw);
    if constexpr (G == 4) radix16_group<IDLE, Q, G, 4>(x, tw);
}
template <int G = FRIEDA_NTT_GROUP, typename TW>
__device__ __forceinline__ void radix16_stage(uint32_t (&x)[16], const TW& tw) {
    if constexpr (FRIEDA_NTT_IDLE != 0) {
        radix16_layer<FRIEDA_NTT_IDLE, 0, G>(x, tw);
        radix16_layer<FRIEDA_NTT_IDLE, 1, G>(x, tw);
        radix16_layer<FRIEDA_NTT_IDLE, 2, G>(x, tw);
        radix16_layer<FRIEDA_NTT_IDLE, 3, G>(x, tw);
        if constexpr (((FRIEDA_NTT_IDLE >> 8) & 3) != 0) {  // back to the priority of everything that is not a butterfly (ntt_enter)
            if constexpr (FRIEDA_NTT_OUTER_PRIO == 0) asm volatile("s_setprio 0" : "+v"(x[0]), "+v"(x[8]), "+v"(x[7]), "+v"(x[15]));
            if constexpr (FRIEDA_NTT_OUTER_PRIO == 1) asm volatile("s_setprio 1" : "+v"(x[0]), "+v"(x[8]), "+v"(x[7]), "+v"(x[15]));
            if constexpr (FRIEDA_NTT_OUTER_PRIO == 2) asm volatile("s_setprio 2" : "+v"(x[0]), "+v"(x[8]), "+v"(x[7]), "+v"(x[15]));
            if constexpr (FRIEDA_NTT_OUTER_PRIO == 3) asm volatile("s_setprio 3" : "+v"(x[0]), "+v"(x[8]), "+v"(x[7]), "+v"(x[15]));
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int bit = 3 - q;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (r & (1 << bit)) continue;
                const int u = r >> (bit + 1);
                const uint32_t t = m31_mul(x[r | (1 << bit)], tw[(1 << q) - 1 + u] >> 1);  // (the callers hand doubled twiddles)
                const uint32_t v = x[r];
                x[r] = m31_add(v, t);
                x[r | (1 << bit)] = m31_sub(v, t);
            }
        }
    }
}

// circle-layer twiddle Y[h] from the first line level: pairs (x, y) -> [y, -y, -x, x]
__device__ __forceinline__ uint32_t circle_twiddle(const uint32_t* __restrict__ tw, uint32_t n, uint32_t h, uint32_t init_y) {
    if (n < 3) return (h & 1u) ? m31_neg(init_y) : init_y;  // n == 1: [y]; n == 2: [y, -y]
    uint32_t j = h >> 2, r = h & 3u;
    uint32_t v = tw[2 * j + (r < 2 ? 1 : 0)];
    return (r == 1 || r == 2) ? m31_neg(v) : v;
}

struct NttArgs {
    const uint32_t* in;
    size_t in_stride;
    uint32_t in_mask;
    uint32_t in_limit;  // source words at (index & in_mask) >= in_limit read as zero (the zero-padded coefficients of explicit padded layers)
    uint32_t* out;
    size_t out_stride;
    const uint32_t* tw;
    uint32_t n, i_hi, i_lo, log_w, init_y;
    uint32_t ncols;       // columns handled by one workgroup (<= 4); grid.y strides over groups of this many
    uint32_t n_stages;    // <= 3
    uint32_t stage_r[3];  // layers per stage, top stage first
    size_t bstride_w;     // batch: words between consecutive blobs' buffers (both in and out); blob = blockIdx.z
    uint32_t rep_log;     // ntt_tile12_rep_kernel: a workgroup produces the tiles of 2^rep_log consecutive high blocks from one read of its source tile
};

// One stage of R layers on tile bits [lo, lo + R): every thread processes groups of 2^R elements.
// LO >= 0 fixes `lo` at compile time (the hot shapes): the 2^R LDS addresses of a group are then pad(base) plus immediates
// (pad(base | x) == pad(base) + pad(x) because base has zeros where x = r << lo has bits) and so are the twiddle addresses.
template <int R, int LO>
__device__ __forceinline__ void run_stage(uint32_t* lds, const NttArgs& a, uint32_t tb, uint32_t lo_rt, uint32_t hblk) {
    constexpr int E = 1 << R;
    const uint32_t lo = LO >= 0 ? (uint32_t)LO : lo_rt;
    const uint32_t n_groups = 1u << (tb - R);
    for (uint32_t g = threadIdx.x; g < n_groups; g += NTT_THREADS) {
        const uint32_t base = ((g >> lo) << (lo + R)) | (g & ((1u << lo) - 1));
        const uint32_t pbase = pad(base);
        // twiddles: layer q of the stage acts on tile bit b = lo + R - 1 - q, global layer i = i_lo + b - log_w
        uint32_t twd[E - 1];
#pragma unroll
        for (int q = 0; q < R; q++) {
            const uint32_t b = lo + R - 1 - q;
            const uint32_t i = a.i_lo + b - a.log_w;
            const uint32_t hbase = (hblk << (a.i_hi - i)) | (base >> (b + 1));  // its low q bits are zero
            if (i >= 1) {
                const uint32_t* lvl = a.tw + tw_level_offset_dev(a.n, i - 1) + hbase;
#pragma unroll
                for (int u = 0; u < (1 << q); u++) twd[(1 << q) - 1 + u] = lvl[u];
            } else {
#pragma unroll
                for (int u = 0; u < (1 << q); u++) twd[(1 << q) - 1 + u] = circle_twiddle(a.tw, a.n, hbase + (uint32_t)u, a.init_y);
            }
        }
        for (uint32_t c = 0; c < a.ncols; c++) {
            uint32_t* col = lds + c * TILE_WORDS + pbase;
            uint32_t x[E];
#pragma unroll
            for (int r = 0; r < E; r++) x[r] = col[pad((uint32_t)r << lo)];
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int bit = R - 1 - q;
#pragma unroll
                for (int r = 0; r < E; r++) {
                    if (r & (1 << bit)) continue;
                    const int u = r >> (bit + 1);
                    const uint32_t t = m31_mul(x[r | (1 << bit)], twd[(1 << q) - 1 + u]);
                    const uint32_t v = x[r];
                    x[r] = m31_add(v, t);
                    x[r | (1 << bit)] = m31_sub(v, t);
                }
            }
#pragma unroll
            for (int r = 0; r < E; r++) col[pad((uint32_t)r << lo)] = x[r];
        }
    }
}

__global__ __launch_bounds__(NTT_THREADS) void ntt_tile_kernel(NttArgs a) {
    extern __shared__ uint32_t lds[];
    const uint32_t t = a.i_hi - a.i_lo + 1;
    const uint32_t tb = t + a.log_w;  // tile bits (<= 12)
    const uint32_t tile = 1u << tb;
    const uint32_t wmask = (1u << a.log_w) - 1;
    const uint32_t nwb_log = a.i_lo - a.log_w;
    const uint32_t wblk = blockIdx.x & ((1u << nwb_log) - 1);
    const uint32_t hblk = blockIdx.x >> nwb_log;
    const uint32_t gbase = (hblk << (a.i_hi + 1)) | (wblk << a.log_w);
    const size_t col0 = (size_t)blockIdx.y * a.ncols;
    const uint32_t* in = a.in + col0 * a.in_stride + blockIdx.z * a.bstride_w;
    uint32_t* out = a.out + col0 * a.out_stride + blockIdx.z * a.bstride_w;
    // 16-byte global accesses need 4-word runs: tiles of >= 4 words whose low run (2^log_w, or the whole tile when
    // log_w == 0) is a multiple of 4, and 16-byte aligned column bases
    const bool vec = tb >= 2 && (a.log_w == 0 || a.log_w >= 2) && ((a.in_stride | a.out_stride) & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(a.in) | reinterpret_cast<uintptr_t>(a.out)) & 15) == 0 && (a.in_mask & 3u) == 3u &&
                     (a.bstride_w & 3) == 0;

    for (uint32_t c = 0; c < a.ncols; c++) {
        uint32_t* col = lds + c * TILE_WORDS;
        const uint32_t* src = in + c * a.in_stride;
        if (vec) {
            for (uint32_t e = 4 * threadIdx.x; e < tile; e += 4 * NTT_THREADS) {
                uint32_t g = gbase | ((e >> a.log_w) << a.i_lo) | (e & wmask);
                uint4 v = *reinterpret_cast<const uint4*>(src + (g & a.in_mask));
                uint32_t p = pad(e);  // e is a multiple of 4: the four words stay inside one 16-word group
                col[p] = v.x;
                col[p + 1] = v.y;
                col[p + 2] = v.z;
                col[p + 3] = v.w;
            }
        } else {
            for (uint32_t e = threadIdx.x; e < tile; e += NTT_THREADS) {
                uint32_t g = gbase | ((e >> a.log_w) << a.i_lo) | (e & wmask);
                col[pad(e)] = src[g & a.in_mask];
            }
        }
    }
    __syncthreads();

    uint32_t top = tb;  // one past the top tile bit of the next stage
    for (uint32_t s = 0; s < a.n_stages; s++) {
        const uint32_t r = a.stage_r[s], lo = top - r;
        if (r == 4 && lo == 8)
            run_stage<4, 8>(lds, a, tb, lo, hblk);
        else if (r == 4 && lo == 4)
            run_stage<4, 4>(lds, a, tb, lo, hblk);
        else if (r == 4 && lo == 0)
            run_stage<4, 0>(lds, a, tb, lo, hblk);
        else if (r == 4)
            run_stage<4, -1>(lds, a, tb, lo, hblk);
        else if (r == 3)
            run_stage<3, -1>(lds, a, tb, lo, hblk);
        else if (r == 2)
            run_stage<2, -1>(lds, a, tb, lo, hblk);
        else
            run_stage<1, -1>(lds, a, tb, lo, hblk);
        top = lo;
        __syncthreads();
    }

    for (uint32_t c = 0; c < a.ncols; c++) {
        const uint32_t* col = lds + c * TILE_WORDS;
        uint32_t* dst = out + c * a.out_stride;
        if (vec) {
            for (uint32_t e = 4 * threadIdx.x; e < tile; e += 4 * NTT_THREADS) {
                uint32_t g = gbase | ((e >> a.log_w) << a.i_lo) | (e & wmask);
                uint32_t p = pad(e);
                *reinterpret_cast<uint4*>(dst + g) = make_uint4(col[p], col[p + 1], col[p + 2], col[p + 3]);
            }
        } else {
            for (uint32_t e = threadIdx.x; e < tile; e += NTT_THREADS) {
                uint32_t g = gbase | ((e >> a.log_w) << a.i_lo) | (e & wmask);
                dst[g] = col[pad(e)];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The hot shape: a full 4096-word tile whose layers split into NS stages of exactly four (NS = 3, LOG_W = 0: the
// contiguous last pass of 12 layers; NS = 2, LOG_W = 4: a strided pass of 8 layers).  One group of 16 elements per thread
// and stage, so all 15 * NS twiddles of the thread are loaded once into registers and the columns are streamed through a
// single 17 KiB LDS tile one after the other (9 workgroups fit a CU's LDS; registers allow 5-6 waves per SIMD).  The next
// column's tile is prefetched into registers while the current one is computed; the contiguous pass stores its results
// straight from registers (16 consecutive words per thread).
// ------------------------------------------------------------------------------------------------
template <int NS, int LOG_W>
__global__ __launch_bounds__(NTT_THREADS) void ntt_tile12_kernel(NttArgs a) {
    ntt_enter();
    __shared__ uint32_t lds[TILE_WORDS];
    const uint32_t g = threadIdx.x;
    const uint32_t nwb_log = a.i_lo - LOG_W;
    const uint32_t wblk = blockIdx.x & ((1u << nwb_log) - 1);
    const uint32_t hblk = blockIdx.x >> nwb_log;
    const uint32_t gbase = (hblk << (a.i_hi + 1)) | (wblk << LOG_W);
    constexpr uint32_t wmask = (1u << LOG_W) - 1;
    const size_t col0 = (size_t)blockIdx.y * a.ncols;
    const uint32_t* in = a.in + col0 * a.in_stride + blockIdx.z * a.bstride_w;
    uint32_t* out = a.out + col0 * a.out_stride + blockIdx.z * a.bstride_w;

    // per-stage group base (padded) and twiddles
    uint32_t pbase[NS];
    uint32_t twd[NS][15];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const uint32_t lo = 8 - 4 * s;
        const uint32_t base = stage_group_base(g, lo);
        pbase[s] = pad(base);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t b = lo + 3 - q;
            const uint32_t i = a.i_lo + b - LOG_W;
            const uint32_t hbase = (hblk << (a.i_hi - i)) | (base >> (b + 1));
            if (LOG_W == 0 && s == NS - 1 && q == 3) {  // i == 0: the circle layer
#pragma unroll
                for (int u = 0; u < 8; u++) twd[s][7 + u] = 2u * circle_twiddle(a.tw, a.n, hbase + (uint32_t)u, a.init_y);
            } else {
                const uint32_t* lvl = a.tw + tw_level_offset_dev(a.n, i - 1) + hbase;
#pragma unroll
                for (int u = 0; u < (1 << q); u++) {
                    uint32_t v = lvl[u];
                    // stage 0 acts on tile bits 8..11: its twiddle index has no thread-dependent bits (base >> (b+1) == 0 for
                    // g < 256), so the 15 values are workgroup-uniform and live in scalar registers
                    if (s == 0) v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
                    twd[s][(1 << q) - 1 + u] = 2u * v;  // doubled: radix16_group
                }
            }
        }
    }

    // tile element e (multiple of 4) of this thread's k-th 16-byte piece, and its global index
    auto piece_e = [&](int kk) { return 4u * g + 1024u * (uint32_t)kk; };
    auto global_of = [&](uint32_t e) { return gbase | ((e >> LOG_W) << a.i_lo) | (e & wmask); };

    // 16 bytes of a source column; indices at or beyond in_limit are the zero padding of the coefficient vector
    auto load4 = [&](const uint32_t* src, uint32_t gidx) {
        const uint32_t idx = gidx & a.in_mask;
        return idx < a.in_limit ? *reinterpret_cast<const uint4*>(src + idx) : make_uint4(0u, 0u, 0u, 0u);
    };
    uint4 pre[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) pre[kk] = load4(in, global_of(piece_e(kk)));

    for (uint32_t c = 0; c < a.ncols; c++) {
        // tile of column c: registers -> LDS
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const uint32_t p = pad(piece_e(kk));
            lds[p] = pre[kk].x;
            lds[p + 1] = pre[kk].y;
            lds[p + 2] = pre[kk].z;
            lds[p + 3] = pre[kk].w;
        }
        __syncthreads();
        if (c + 1 < a.ncols) {  // prefetch the next column while this one is computed
            const uint32_t* src = in + (size_t)(c + 1) * a.in_stride;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) pre[kk] = load4(src, global_of(piece_e(kk)));
        }
        uint32_t x[16];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const uint32_t lo = 8 - 4 * s;
            uint32_t* col = lds + pbase[s];
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = col[pad((uint32_t)r << lo)];
            radix16_stage(x, twd[s]);
            if (s + 1 < NS || LOG_W != 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) col[pad((uint32_t)r << lo)] = x[r];
            }
            // The stages on tile bits 4..7 and 0..3 only exchange data inside a wave's own 1024 elements (wave w = threads 64 w ..
            // 64 w + 63 owns elements [1024 w, 1024 w + 1024) in both), and the LDS serves a wave's accesses in order: between those
            // two stages no workgroup barrier is needed, only that the compiler keeps the order.
            if (LOG_W == 0 && NS == 3 && s == 1)
                __builtin_amdgcn_wave_barrier();
            else
                __syncthreads();
        }
        uint32_t* dst = out + (size_t)c * a.out_stride;
        if (LOG_W == 0) {
            // last stage has lo = 0: the thread holds tile elements 16 g .. 16 g + 15, contiguous in memory
            uint4* o = reinterpret_cast<uint4*>(dst + gbase + 16u * g);
            o[0] = make_uint4(x[0], x[1], x[2], x[3]);
            o[1] = make_uint4(x[4], x[5], x[6], x[7]);
            o[2] = make_uint4(x[8], x[9], x[10], x[11]);
            o[3] = make_uint4(x[12], x[13], x[14], x[15]);
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const uint32_t e = piece_e(kk), p = pad(e);
                *reinterpret_cast<uint4*>(dst + global_of(e)) = make_uint4(lds[p], lds[p + 1], lds[p + 2], lds[p + 3]);
            }
            __syncthreads();  // the next column overwrites the tile
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The strided 8-layer pass when it reads the coefficient vector itself (the first pass): the zero-padded layers above it only
// replicate the coefficients, so the workgroups of all 2^(n - 1 - i_hi) high blocks read the SAME source tile and differ in their
// twiddles only.  With one workgroup per high block every coefficient was fetched once per block (208 MB from the memory side for
// 16.8 MB of coefficients at n = 24, profiles/r02_prove24_traffic.json).  Here a workgroup reads its source tile of COLS columns
// once into registers (16 words per thread and column) and loops over 2^rep_log high blocks: twiddles of the block, then per
// column registers -> LDS -> two radix-16 stages -> store.  Twiddles are still shared by the workgroup's columns.
// ------------------------------------------------------------------------------------------------
template <int COLS>
__global__ __launch_bounds__(NTT_THREADS) void ntt_tile12_rep_kernel(NttArgs a) {
    ntt_enter();
    constexpr int NS = 2;
    constexpr uint32_t LOG_W = MID_LOG_W;
    __shared__ uint32_t lds[TILE_WORDS];
    const uint32_t g = threadIdx.x;
    const uint32_t nwb_log = a.i_lo - LOG_W;
    const uint32_t wblk = blockIdx.x & ((1u << nwb_log) - 1);
    const uint32_t hblk0 = (blockIdx.x >> nwb_log) << a.rep_log;
    constexpr uint32_t wmask = (1u << LOG_W) - 1;
    const size_t col0 = (size_t)blockIdx.y * COLS;
    const uint32_t* in = a.in + col0 * a.in_stride + blockIdx.z * a.bstride_w;
    uint32_t* out = a.out + col0 * a.out_stride + blockIdx.z * a.bstride_w;

    auto piece_e = [&](int kk) { return 4u * g + 1024u * (uint32_t)kk; };
    // tile element e -> index below the high block (the source index is this, masked: independent of the high block)
    auto low_of = [&](uint32_t e) { return (wblk << LOG_W) | ((e >> LOG_W) << a.i_lo) | (e & wmask); };

    uint4 src[COLS][4];
#pragma unroll
    for (int c = 0; c < COLS; c++) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const uint32_t idx = low_of(piece_e(kk)) & a.in_mask;
            src[c][kk] = idx < a.in_limit ? *reinterpret_cast<const uint4*>(in + (size_t)c * a.in_stride + idx) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    uint32_t pbase[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const uint32_t lo = 8 - 4 * s;
        pbase[s] = pad(stage_group_base(g, lo));
    }

#pragma unroll 1
    for (uint32_t h = 0; h < (1u << a.rep_log); h++) {
        const uint32_t hblk = hblk0 + h;
        const uint32_t gbase = hblk << (a.i_hi + 1);
        uint32_t twd[NS][15];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const uint32_t lo = 8 - 4 * s;
            const uint32_t base = stage_group_base(g, lo);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t b = lo + 3 - q;
                const uint32_t i = a.i_lo + b - LOG_W;  // >= i_lo >= 12: always a line layer
                const uint32_t hbase = (hblk << (a.i_hi - i)) | (base >> (b + 1));
                const uint32_t* lvl = a.tw + tw_level_offset_dev(a.n, i - 1) + hbase;
#pragma unroll
                for (int u = 0; u < (1 << q); u++) {
                    uint32_t v = lvl[u];
                    if (s == 0) v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);  // workgroup-uniform (ntt_tile12_kernel)
                    twd[s][(1 << q) - 1 + u] = 2u * v;  // doubled: radix16_group
                }
            }
        }
#pragma unroll
        for (int c = 0; c < COLS; c++) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const uint32_t p = pad(piece_e(kk));
                lds[p] = src[c][kk].x;
                lds[p + 1] = src[c][kk].y;
                lds[p + 2] = src[c][kk].z;
                lds[p + 3] = src[c][kk].w;
            }
            __syncthreads();
            uint32_t x[16];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const uint32_t lo = 8 - 4 * s;
                uint32_t* col = lds + pbase[s];
#pragma unroll
                for (int r = 0; r < 16; r++) x[r] = col[pad((uint32_t)r << lo)];
                radix16_stage(x, twd[s]);
#pragma unroll
                for (int r = 0; r < 16; r++) col[pad((uint32_t)r << lo)] = x[r];
                __syncthreads();
            }
            uint32_t* dst = out + (size_t)c * a.out_stride + gbase;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const uint32_t e = piece_e(kk), p = pad(e);
                *reinterpret_cast<uint4*>(dst + low_of(e)) = make_uint4(lds[p], lds[p + 1], lds[p + 2], lds[p + 3]);
            }
            __syncthreads();  // the next tile overwrites the LDS buffer
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The last transform pass fused with leaf hashing (commit(): src/commit.rs:16-21; FriProver::commit_first_layer: src/proof.rs:48-52)
// ------------------------------------------------------------------------------------------------
// A workgroup of the contiguous last pass ends up holding the same 4096 points of all four coordinate columns — exactly the
// 4096 leaves (4 column words each) of one subtree of the first Merkle tree.  Instead of writing the evaluation and having a
// tree kernel read it straight back, the thread keeps its 16 consecutive points of every column in registers (it already holds
// them after the last radix-16 stage), hashes the 16 leaves and the 15 nodes above them, and the workgroup adds two more
// levels through LDS: seven tree levels (2^n leaves .. 2^(n-6) nodes) leave the kernel, 64 hashes per workgroup at the top.
// The evaluation is written once (generate_proof folds and opens it) or not at all (commit() needs only the root); it is
// never re-read.  The transform's memory and LDS phases of one workgroup run under the hashing of the others on the CU.
// Hashing runs four leaves at a time in a rolled loop (the 7-compression body of tree5r, ~54 KB of code) over register arrays
// that rotate by four each iteration, so that every index is static.
struct NttTreeArgs {
    NttArgs a;
    uint8_t* layers;    // STORE_ALL: tree storage (leaves-first offsets); the leaf level itself is never written (nothing reads it)
    uint8_t* last_out;  // !STORE_ALL: the 2^(n-6) hashes of the last level produced
    size_t bstride;     // batch: bytes between blobs' workspaces (blob = blockIdx.z)
};


// The contiguous last pass (layers 11 .. 0, one 4096-word tile per column) on all four coordinate columns of workgroup `hblk`, the
// results kept in registers: v[c][r] = column c at tile element 16 g + r (global index (hblk << 12) + 16 g + r).  Shared by the
// kernel that hashes them (ntt_last_tree_kernel) and the one that folds them (ntt_last_fold_kernel).
template <bool STORE_ALL>
__device__ __forceinline__ void last_pass_four_columns(const NttArgs& a, uint32_t* lds, uint32_t g, uint32_t hblk, const uint32_t* in, uint32_t* out,
                                                       uint32_t (&v)[4][16]) {
    const uint32_t gbase = hblk << TILE_LOG;
    uint32_t pbase[3];
    uint32_t twd[3][15];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const uint32_t lo = 8 - 4 * s;
        const uint32_t base = stage_group_base(g, lo);
        pbase[s] = pad(base);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t b = lo + 3 - q;  // tile bit == global layer index
            const uint32_t hbase = (hblk << (11 - b)) | (base >> (b + 1));
            if (s == 2 && q == 3) {  // layer 0: the circle layer
#pragma unroll
                for (int u = 0; u < 8; u++) twd[s][7 + u] = 2u * circle_twiddle(a.tw, a.n, hbase + (uint32_t)u, a.init_y);
            } else {
                const uint32_t* lvl = a.tw + tw_level_offset_dev(a.n, b - 1) + hbase;
#pragma unroll
                for (int u = 0; u < (1 << q); u++) {
                    uint32_t v = lvl[u];
                    if (s == 0) v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);  // workgroup-uniform (ntt_tile12_kernel)
                    twd[s][(1 << q) - 1 + u] = 2u * v;  // doubled: radix16_group
                }
            }
        }
    }
    auto piece_e = [&](int kk) { return 4u * g + 1024u * (uint32_t)kk; };

    uint4 pre[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) pre[kk] = *reinterpret_cast<const uint4*>(in + ((gbase | piece_e(kk)) & a.in_mask));

#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const uint32_t p = pad(piece_e(kk));
            lds[p] = pre[kk].x;
            lds[p + 1] = pre[kk].y;
            lds[p + 2] = pre[kk].z;
            lds[p + 3] = pre[kk].w;
        }
        __syncthreads();
        if (c + 1 < 4) {
            const uint32_t* src = in + (size_t)(c + 1) * a.in_stride;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) pre[kk] = *reinterpret_cast<const uint4*>(src + ((gbase | piece_e(kk)) & a.in_mask));
        }
#pragma unroll
        for (int s = 0; s < 3; s++) {
            const uint32_t lo = 8 - 4 * s;
            uint32_t* col = lds + pbase[s];
            uint32_t* x = v[c];
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = col[pad((uint32_t)r << lo)];
            radix16_stage<FRIEDA_NTT_GROUP_FUSED>(*reinterpret_cast<uint32_t(*)[16]>(x), twd[s]);
            if (s < 2) {
#pragma unroll
                for (int r = 0; r < 16; r++) col[pad((uint32_t)r << lo)] = x[r];
            }
            if (s == 1)
                __builtin_amdgcn_wave_barrier();  // stages 1 and 2 are wave-local (ntt_tile12_kernel)
            else
                __syncthreads();
        }
        if (STORE_ALL) {  // generate_proof folds and opens the evaluation; commit() never reads it
            uint4* o = reinterpret_cast<uint4*>(out + (size_t)c * a.out_stride + gbase + 16u * g);
            o[0] = make_uint4(v[c][0], v[c][1], v[c][2], v[c][3]);
            o[1] = make_uint4(v[c][4], v[c][5], v[c][6], v[c][7]);
            o[2] = make_uint4(v[c][8], v[c][9], v[c][10], v[c][11]);
            o[3] = make_uint4(v[c][12], v[c][13], v[c][14], v[c][15]);
        }
    }

}

FR_CLOCK_DECL(g_clock_ntt_last_tree)
// REG_ONLY: stop after the five register levels (256 nodes of level n - 4 per workgroup): see tree5r_kernel in tree.hip
// TP: the compressions in the throughput form (launches of >= 768 tiles; below that the launch is a latency chain: tree.hip tp_launch)
template <bool STORE_ALL, bool REG_ONLY, bool TP>
__global__ __launch_bounds__(NTT_THREADS) void ntt_last_tree_kernel(NttTreeArgs A) {
    ntt_enter();
    __shared__ uint32_t lds[TILE_WORDS];
    __shared__ __attribute__((aligned(16))) uint32_t RC[8 * (256 + 4)];
    __shared__ __attribute__((aligned(16))) uint32_t RD[8 * (128 + 4)];
    const NttArgs& a = A.a;
    uint32_t g_tie = threadIdx.x;
    FR_CLOCK_BEGIN(g_tie)
    const uint32_t g = g_tie;
    const uint32_t hblk = blockIdx.x;  // i_hi == 11, i_lo == 0, log_w == 0: one contiguous 4096-word tile per workgroup
    const uint32_t gbase = hblk << TILE_LOG;
    const uint32_t* in = a.in + blockIdx.z * a.bstride_w;
    uint32_t* out = a.out + blockIdx.z * a.bstride_w;

    uint32_t v[4][16];  // the thread's 16 consecutive points (tile elements 16 g .. 16 g + 15) of every column
    last_pass_four_columns<STORE_ALL>(a, lds, g, hblk, in, out, v);

    // ---- 16 leaves -> 1 node of level n - 4, in registers ----
    const uint32_t m = a.n;
    uint8_t* layers = A.layers + blockIdx.z * A.bstride;
    uint8_t* out_b = STORE_ALL ? layers + layer_off(m, m - 1) : nullptr;
    uint8_t* out_c = STORE_ALL ? layers + layer_off(m, m - 2) : nullptr;
    uint8_t* out_d = STORE_ALL ? layers + layer_off(m, m - 3) : nullptr;
    uint8_t* out_e = STORE_ALL ? layers + layer_off(m, m - 4) : (REG_ONLY ? A.last_out + blockIdx.z * A.bstride : nullptr);
    const size_t leaf0 = (size_t)gbase + 16u * g;
    uint32_t hprev[8], hdprev[8];
#pragma unroll 1
    for (int it = 0; it < 4; it++) {
        const size_t l0 = leaf0 + 4u * (uint32_t)it;
        uint32_t hb[2][8];
        // (written out through a lambda: the unroller refuses loops whose body carries the compressions' inline asm)
        auto half_ab = [&](auto half_c) {
            constexpr int half = decltype(half_c)::value;
            uint32_t ha[2][8];
            leaf_hash<TP ? FRIEDA_B2_IDLE_NTT_LEAF : B2_LAT>(v[0][2 * half], v[1][2 * half], v[2][2 * half], v[3][2 * half], ha[0]);
            leaf_hash<TP ? FRIEDA_B2_IDLE_NTT_LEAF : B2_LAT>(v[0][2 * half + 1], v[1][2 * half + 1], v[2][2 * half + 1], v[3][2 * half + 1], ha[1]);
            uint32_t mm[16];
#pragma unroll
            for (int w = 0; w < 8; w++) mm[w] = ha[0][w], mm[8 + w] = ha[1][w];
            b2_merkle_block<TP ? FRIEDA_B2_IDLE_NTT_NODE : B2_LAT>(mm, hb[half]);
        };
        half_ab(std::integral_constant<int, 0>{});
        half_ab(std::integral_constant<int, 1>{});
        if (STORE_ALL) {
            store_hash(out_b, l0 >> 1, hb[0]);
            store_hash(out_b, (l0 >> 1) + 1, hb[1]);
        }
        uint32_t hc[8];
        {
            uint32_t mm[16];
#pragma unroll
            for (int w = 0; w < 8; w++) mm[w] = hb[0][w], mm[8 + w] = hb[1][w];
            b2_merkle_block<TP ? FRIEDA_B2_IDLE_NTT_NODE : B2_LAT>(mm, hc);
        }
        if (STORE_ALL) store_hash(out_c, l0 >> 2, hc);
        if ((it & 1) == 0) {
#pragma unroll
            for (int w = 0; w < 8; w++) hprev[w] = hc[w];
        } else {
            uint32_t hd[8], mm[16];
#pragma unroll
            for (int w = 0; w < 8; w++) mm[w] = hprev[w], mm[8 + w] = hc[w];
            b2_merkle_block<TP ? FRIEDA_B2_IDLE_NTT_NODE : B2_LAT>(mm, hd);
            if (STORE_ALL) store_hash(out_d, l0 >> 3, hd);
            if (it == 1) {
#pragma unroll
                for (int w = 0; w < 8; w++) hdprev[w] = hd[w];
            } else {
                uint32_t he[8];
#pragma unroll
                for (int w = 0; w < 8; w++) mm[w] = hdprev[w], mm[8 + w] = hd[w];
                b2_merkle_block<TP ? FRIEDA_B2_IDLE_NTT_NODE : B2_LAT>(mm, he);
                if (STORE_ALL || REG_ONLY) store_hash(out_e, l0 >> 4, he);
                if (!REG_ONLY) lds_put(RC, 256 + 4, g, he);
            }
        }
        // the next four leaves move to the front of the arrays (static indices only)
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
            for (int r = 0; r < 12; r++) v[c][r] = v[c][r + 4];
        }
    }
    if (REG_ONLY) return;
    __syncthreads();
    // ---- levels n - 5 and n - 6 through LDS (as tree5r) ----
    const size_t wg_e = (size_t)gbase >> 4;  // index of this workgroup's first node at level n - 4
    if (g < 128) {
        uint32_t mm[16], h[8];
        lds_children(RC, 256 + 4, g, mm);
        b2_merkle_block<TP ? FRIEDA_B2_IDLE_NTT_NODE : B2_LAT>(mm, h);
        if (STORE_ALL) store_hash(layers + layer_off(m, m - 5), (wg_e >> 1) + g, h);
        lds_put(RD, 128 + 4, g, h);
    }
    __syncthreads();
    if (g < 64) {
        uint32_t mm[16], h[8];
        lds_children(RD, 128 + 4, g, mm);
        b2_merkle_block<TP ? FRIEDA_B2_IDLE_NTT_NODE : B2_LAT>(mm, h);
        uint8_t* dst = STORE_ALL ? layers + layer_off(m, m - 6) : A.last_out + blockIdx.z * A.bstride;
        store_hash(dst, (wg_e >> 2) + g, h);
        FR_CLOCK_END(g_clock_ntt_last_tree, h[0])
    }
}

// pure replication (L == 0: a constant polynomial has no real layers)
__global__ void ntt_broadcast_kernel(const uint32_t* __restrict__ in, size_t in_stride, uint32_t* __restrict__ out,
                                     size_t out_stride, size_t n_out, size_t bstride_w) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_out) out[blockIdx.z * bstride_w + (size_t)blockIdx.y * out_stride + i] = in[blockIdx.z * bstride_w + (size_t)blockIdx.y * in_stride];
}

void set_stages(NttArgs& a, uint32_t t) {
    // split t layers into ceil(t / 4) stages of nearly equal size, larger first
    uint32_t ns = (t + 3) / 4;
    a.n_stages = ns;
    uint32_t left = t;
    for (uint32_t s = 0; s < ns; s++) {
        uint32_t r = (left + (ns - s) - 1) / (ns - s);
        a.stage_r[s] = r;
        left -= r;
    }
}

// ------------------------------------------------------------------------------------------------
// The last transform pass fused with the first FRI fold round (Level B: frieda_circle_evaluate_fold2)
// ------------------------------------------------------------------------------------------------
// The same workgroup-holds-4096-points-of-all-four-coordinates situation, folded instead of hashed: a thread's 16 consecutive points
// are 8 pairs of fold_circle_into_line, whose 8 results are 4 pairs of the first fold_line (stwo backend/cpu/fri.rs; the pairs
// (2i, 2i + 1) of the bit-reversed order are adjacent).  The evaluation, the first line layer and the second leave in one pass over
// the tile: 16-byte stores only, nothing re-read.  ACC: FriOps::fold_circle_into_line's accumulate form, dst = dst * alpha^2 + fold.
struct NttFoldArgs {
    NttArgs a;
    const uint32_t* itw;
    uint32_t inv_init_y;
    uint32_t alpha0[4], alpha1[4];
    uint32_t* line1;  // 4 coordinate columns of 2^(n-1) words
    uint32_t* line2;  // 4 coordinate columns of 2^(n-2) words
};

template <bool ACC>
__global__ __launch_bounds__(NTT_THREADS) void ntt_last_fold_kernel(NttFoldArgs A) {
    ntt_enter();
    __shared__ uint32_t lds[TILE_WORDS];
    const NttArgs& a = A.a;
    const uint32_t g = threadIdx.x, hblk = blockIdx.x;
    uint32_t v[4][16];
    last_pass_four_columns<true>(a, lds, g, hblk, a.in, a.out, v);
    const uint32_t n = a.n;
    const size_t e0 = ((size_t)hblk << TILE_LOG) + 16u * g;  // this thread's first point
    const QM31Mat am0 = qm_matrix({A.alpha0[0], A.alpha0[1], A.alpha0[2], A.alpha0[3]});
    const QM31Mat am1 = qm_matrix({A.alpha1[0], A.alpha1[1], A.alpha1[2], A.alpha1[3]});
    // inverse twiddles: circle pairs i = e0 / 2 + k read inverse-Y[i] = [iy, -iy, -ix, ix] from inverse level 0 (two (x, y) pairs
    // cover the 8); line pairs j = e0 / 4 + k read inverse level 0 at j (the same four words)
    const uint4 t0 = *reinterpret_cast<const uint4*>(A.itw + (e0 >> 2));
    const uint32_t ix[2] = {t0.x, t0.z}, iy[2] = {t0.y, t0.w};
    const uint32_t il[4] = {t0.x, t0.y, t0.z, t0.w};
    const size_t s1 = (size_t)1 << (n - 1), s2 = (size_t)1 << (n - 2);
    QM31 l1[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int pr = k >> 2, r = k & 3;
        const uint32_t it = r == 0 ? iy[pr] : (r == 1 ? m31_neg(iy[pr]) : (r == 2 ? m31_neg(ix[pr]) : ix[pr]));
        l1[k] = qm_fold_pair(QM31{v[0][2 * k], v[1][2 * k], v[2][2 * k], v[3][2 * k]},
                             QM31{v[0][2 * k + 1], v[1][2 * k + 1], v[2][2 * k + 1], v[3][2 * k + 1]}, it, am0);
    }
    uint32_t* d1 = A.line1 + (e0 >> 1);
    if (ACC) {
        const QM31 al = {A.alpha0[0], A.alpha0[1], A.alpha0[2], A.alpha0[3]};
        const QM31 asq = qm_mul(al, al);
        uint4 o[4][2];
#pragma unroll
        for (int c = 0; c < 4; c++) o[c][0] = reinterpret_cast<const uint4*>(d1 + c * s1)[0], o[c][1] = reinterpret_cast<const uint4*>(d1 + c * s1)[1];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            auto w = [&](int c) { const uint4 q4 = o[c][k >> 2]; return (k & 3) == 0 ? q4.x : ((k & 3) == 1 ? q4.y : ((k & 3) == 2 ? q4.z : q4.w)); };
            l1[k] = qm_add(qm_mul(QM31{w(0), w(1), w(2), w(3)}, asq), l1[k]);
        }
    }
    {
        uint4* p0 = reinterpret_cast<uint4*>(d1);
        uint4* p1 = reinterpret_cast<uint4*>(d1 + s1);
        uint4* p2 = reinterpret_cast<uint4*>(d1 + 2 * s1);
        uint4* p3 = reinterpret_cast<uint4*>(d1 + 3 * s1);
        p0[0] = make_uint4(l1[0].a, l1[1].a, l1[2].a, l1[3].a), p0[1] = make_uint4(l1[4].a, l1[5].a, l1[6].a, l1[7].a);
        p1[0] = make_uint4(l1[0].b, l1[1].b, l1[2].b, l1[3].b), p1[1] = make_uint4(l1[4].b, l1[5].b, l1[6].b, l1[7].b);
        p2[0] = make_uint4(l1[0].c, l1[1].c, l1[2].c, l1[3].c), p2[1] = make_uint4(l1[4].c, l1[5].c, l1[6].c, l1[7].c);
        p3[0] = make_uint4(l1[0].d, l1[1].d, l1[2].d, l1[3].d), p3[1] = make_uint4(l1[4].d, l1[5].d, l1[6].d, l1[7].d);
    }
    QM31 l2[4];
#pragma unroll
    for (int k = 0; k < 4; k++) l2[k] = qm_fold_pair(l1[2 * k], l1[2 * k + 1], il[k], am1);
    uint32_t* d2 = A.line2 + (e0 >> 2);
    *reinterpret_cast<uint4*>(d2) = make_uint4(l2[0].a, l2[1].a, l2[2].a, l2[3].a);
    *reinterpret_cast<uint4*>(d2 + s2) = make_uint4(l2[0].b, l2[1].b, l2[2].b, l2[3].b);
    *reinterpret_cast<uint4*>(d2 + 2 * s2) = make_uint4(l2[0].c, l2[1].c, l2[2].c, l2[3].c);
    *reinterpret_cast<uint4*>(d2 + 3 * s2) = make_uint4(l2[0].d, l2[1].d, l2[2].d, l2[3].d);
}

// ------------------------------------------------------------------------------------------------
// The same pass for SMALL launches (fewer tiles than the chip has CUs twice over): the four columns side by side
// ------------------------------------------------------------------------------------------------
// With one 256-thread workgroup per tile a 2^20 domain is 256 workgroups = one wave per SIMD, and the launch is a latency chain:
// four columns x three LDS stages one after the other, then the folds (25 us for BASELINE configs[1]).  Here a workgroup has 1024
// threads: quarter c transforms column c in its own LDS tile (4 x 17 KB of dynamic LDS), all at once; the results go back into the
// tiles, and every thread folds 4 consecutive points of all four columns (two circle pairs -> one line pair -> one point of line 2).
constexpr int NTT_CP_THREADS = 1024;
template <bool ACC>
__global__ __launch_bounds__(NTT_CP_THREADS) void ntt_last_fold_cp_kernel(NttFoldArgs A) {
    ntt_enter();
    extern __shared__ uint32_t cp_lds[];
    const NttArgs& a = A.a;
    const uint32_t c = threadIdx.x >> 8, g = threadIdx.x & 255u, hblk = blockIdx.x;
    const uint32_t gbase = hblk << TILE_LOG;
    uint32_t* lds = cp_lds + c * TILE_WORDS;
    const uint32_t* in = a.in + (size_t)c * a.in_stride;
    // ---- column c: 12 layers in three radix-16 stages (as last_pass_four_columns, one column); every global load of the chain —
    // the tile and the three stages' twiddles — is issued before the first wait ----
    uint4 pre[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) pre[kk] = *reinterpret_cast<const uint4*>(in + ((gbase | (4u * g + 1024u * (uint32_t)kk)) & a.in_mask));
    uint32_t twd[3][15], pbase[3];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const uint32_t lo = 8 - 4 * s;
        const uint32_t base = stage_group_base(g, lo);
        pbase[s] = pad(base);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t b = lo + 3 - q;  // tile bit == global layer index
            const uint32_t hbase = (hblk << (11 - b)) | (base >> (b + 1));
            if (s == 2 && q == 3) {  // layer 0: the circle layer
#pragma unroll
                for (int u = 0; u < 8; u++) twd[s][7 + u] = 2u * circle_twiddle(a.tw, a.n, hbase + (uint32_t)u, a.init_y);
            } else {
                const uint32_t* lvl = a.tw + tw_level_offset_dev(a.n, b - 1) + hbase;
#pragma unroll
                for (int u = 0; u < (1 << q); u++) twd[s][(1 << q) - 1 + u] = 2u * lvl[u];  // doubled: radix16_group
            }
        }
    }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const uint32_t p = pad(4u * g + 1024u * (uint32_t)kk);
        lds[p] = pre[kk].x;
        lds[p + 1] = pre[kk].y;
        lds[p + 2] = pre[kk].z;
        lds[p + 3] = pre[kk].w;
    }
    __syncthreads();
    uint32_t x[16];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const uint32_t lo = 8 - 4 * s;
        uint32_t* col = lds + pbase[s];
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = col[pad((uint32_t)r << lo)];
        radix16_stage<FRIEDA_NTT_GROUP_FUSED>(x, twd[s]);
        if (s < 2) {
#pragma unroll
            for (int r = 0; r < 16; r++) col[pad((uint32_t)r << lo)] = x[r];
        }
        if (s == 1)
            __builtin_amdgcn_wave_barrier();  // stages 1 and 2 are wave-local (ntt_tile12_kernel)
        else if (s == 0)
            __syncthreads();
    }
    {  // the evaluation leaves from registers (16 consecutive words per thread); the tile gets it too, for the folds
        uint4* o = reinterpret_cast<uint4*>(a.out + (size_t)c * a.out_stride + gbase + 16u * g);
        o[0] = make_uint4(x[0], x[1], x[2], x[3]);
        o[1] = make_uint4(x[4], x[5], x[6], x[7]);
        o[2] = make_uint4(x[8], x[9], x[10], x[11]);
        o[3] = make_uint4(x[12], x[13], x[14], x[15]);
        uint32_t* mine = lds + pad(16u * g);  // 16 g .. 16 g + 15 sit in one padded group
#pragma unroll
        for (int r = 0; r < 16; r++) mine[r] = x[r];
    }
    __syncthreads();
    // ---- folds: thread T owns tile points 4 T .. 4 T + 3 of all four columns ----
    const uint32_t T = threadIdx.x, n = a.n;
    const size_t e0 = (size_t)gbase + 4u * T;
    uint32_t v[4][4];
#pragma unroll
    for (int cc = 0; cc < 4; cc++) {
        const uint32_t* t4 = cp_lds + cc * TILE_WORDS + pad(4u * T);  // (4 T .. 4 T + 3 do not cross a 16-word group)
#pragma unroll
        for (int k = 0; k < 4; k++) v[cc][k] = t4[k];
    }
    const QM31Mat am0 = qm_matrix({A.alpha0[0], A.alpha0[1], A.alpha0[2], A.alpha0[3]});
    const QM31Mat am1 = qm_matrix({A.alpha1[0], A.alpha1[1], A.alpha1[2], A.alpha1[3]});
    // circle pairs i = e0 / 2, e0 / 2 + 1: inverse-Y[i] = [iy, -iy, -ix, ix] of the inverse level-0 pair (x, y) at index i >> 2;
    // the line pair j = e0 / 4 reads inverse level 0 at j
    const size_t i0 = e0 >> 1;
    const uint2 xy = *reinterpret_cast<const uint2*>(A.itw + 2 * (i0 >> 2));
    const uint32_t it0 = (i0 & 2) ? m31_neg(xy.x) : xy.y, it1 = (i0 & 2) ? xy.x : m31_neg(xy.y);
    const uint32_t il = A.itw[e0 >> 2];
    QM31 l1[2];
    l1[0] = qm_fold_pair(QM31{v[0][0], v[1][0], v[2][0], v[3][0]}, QM31{v[0][1], v[1][1], v[2][1], v[3][1]}, it0, am0);
    l1[1] = qm_fold_pair(QM31{v[0][2], v[1][2], v[2][2], v[3][2]}, QM31{v[0][3], v[1][3], v[2][3], v[3][3]}, it1, am0);
    const size_t s1 = (size_t)1 << (n - 1), s2 = (size_t)1 << (n - 2);
    uint32_t* d1 = A.line1 + i0;
    if (ACC) {
        const QM31 al = {A.alpha0[0], A.alpha0[1], A.alpha0[2], A.alpha0[3]};
        const QM31 asq = qm_mul(al, al);
        uint2 o[4];
#pragma unroll
        for (int cc = 0; cc < 4; cc++) o[cc] = *reinterpret_cast<const uint2*>(d1 + cc * s1);
        l1[0] = qm_add(qm_mul(QM31{o[0].x, o[1].x, o[2].x, o[3].x}, asq), l1[0]);
        l1[1] = qm_add(qm_mul(QM31{o[0].y, o[1].y, o[2].y, o[3].y}, asq), l1[1]);
    }
    *reinterpret_cast<uint2*>(d1) = make_uint2(l1[0].a, l1[1].a);
    *reinterpret_cast<uint2*>(d1 + s1) = make_uint2(l1[0].b, l1[1].b);
    *reinterpret_cast<uint2*>(d1 + 2 * s1) = make_uint2(l1[0].c, l1[1].c);
    *reinterpret_cast<uint2*>(d1 + 3 * s1) = make_uint2(l1[0].d, l1[1].d);
    const QM31 l2 = qm_fold_pair(l1[0], l1[1], il, am1);
    uint32_t* d2 = A.line2 + (e0 >> 2);
    d2[0] = l2.a;
    d2[s2] = l2.b;
    d2[2 * s2] = l2.c;
    d2[3 * s2] = l2.d;
}

}  // namespace

FR_CLOCK_READER(frieda_debug_clock_ntt_last_tree, g_clock_ntt_last_tree)

namespace {
uint32_t evaluate_plan(const Launch& L_, const uint32_t* d_coef, size_t coef_stride, uint32_t ncols, uint32_t L, uint32_t n, uint32_t out_log,
                       const uint32_t* d_tw, DomainScalars ds, uint32_t* d_out, size_t out_stride, const EncodeTreeSink* sink,
                       const EncodeFoldSink* fsink = nullptr);
}

void circle_evaluate(const Launch& L_, const uint32_t* d_coef, size_t coef_stride, uint32_t ncols, uint32_t L, uint32_t n,
                     const uint32_t* d_tw, DomainScalars ds, uint32_t* d_out, size_t out_stride) {
    (void)evaluate_plan(L_, d_coef, coef_stride, ncols, L, n, n, d_tw, ds, d_out, out_stride, nullptr);
}

void circle_evaluate_prefix(const Launch& L_, const uint32_t* d_coef, size_t coef_stride, uint32_t ncols, uint32_t L, uint32_t n, uint32_t out_log,
                            const uint32_t* d_tw, DomainScalars ds, uint32_t* d_out, size_t out_stride) {
    (void)evaluate_plan(L_, d_coef, coef_stride, ncols, L, n, out_log < L ? L : (out_log > n ? n : out_log), d_tw, ds, d_out, out_stride, nullptr);
}

uint32_t circle_evaluate_into_tree(const Launch& L_, const uint32_t* d_coef, size_t coef_stride, uint32_t ncols, uint32_t L, uint32_t n,
                               const uint32_t* d_tw, DomainScalars ds, uint32_t* d_out, size_t out_stride, const EncodeTreeSink* sink) {
    return evaluate_plan(L_, d_coef, coef_stride, ncols, L, n, n, d_tw, ds, d_out, out_stride, sink);
}

hipError_t ntt_opt_in_dynamic_lds() {
    // The opt-in belongs to the CURRENT device's function object: called once per context at creation (frieda_multi drives several
    // devices from one process), so no process-wide flag is kept.
    const int bytes = (int)(MAX_COLS_PER_WG * TILE_WORDS * sizeof(uint32_t));
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_last_fold_cp_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_last_fold_cp_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

bool circle_evaluate_fold2(const Launch& L_, const uint32_t* d_coef, size_t coef_stride, uint32_t L, uint32_t n, const uint32_t* d_tw,
                           DomainScalars ds, uint32_t* d_out, size_t out_stride, const EncodeFoldSink& fs, hipError_t* err) {
    *err = hipSuccess;
    if (evaluate_plan(L_, d_coef, coef_stride, 4, L, n, n, d_tw, ds, d_out, out_stride, nullptr, &fs)) return true;
    // shapes the fused pass does not take: the three operations as they are
    if (!fs.accumulate) {  // (a failed memset would have the fold accumulate onto garbage)
        *err = hipMemsetAsync(fs.line1, 0, sizeof(uint32_t) * ((size_t)4 << (n - 1)), L_.stream);
        if (*err != hipSuccess) return false;
    }
    fold_circle_into_line(L_, fs.line1, (size_t)1 << (n - 1), d_out, out_stride, n, fs.itw, ds, Alpha{{fs.alpha0[0], fs.alpha0[1], fs.alpha0[2], fs.alpha0[3]}});
    fold_line(L_, fs.line1, (size_t)1 << (n - 1), n - 1, n, fs.itw, ds, Alpha{{fs.alpha1[0], fs.alpha1[1], fs.alpha1[2], fs.alpha1[3]}}, fs.line2,
              (size_t)1 << (n - 2));
    return false;
}

namespace {
// `out_log` (L <= out_log <= n): only the first 2^out_log entries of the bit-reversed evaluation are produced.  They depend on all the
// coefficients through the real layers alone (a workgroup's index bits above its layers select its twiddles: the first entries are the
// workgroups whose high index bits are zero), so the same passes run on a shorter grid.  N below is that number of outputs; the domain
// (twiddle tables, index arithmetic inside the kernels) stays 2^n.
// `fsink`: the last pass also folds (ntt_last_fold_kernel) when the shape allows; the return value is then 1, else 0.
uint32_t evaluate_plan(const Launch& L_, const uint32_t* d_coef, size_t coef_stride, uint32_t ncols, uint32_t L, uint32_t n, uint32_t out_log,
                       const uint32_t* d_tw, DomainScalars ds, uint32_t* d_out, size_t out_stride, const EncodeTreeSink* sink,
                       const EncodeFoldSink* fsink) {
    const size_t N = (size_t)1 << out_log;
    hipStream_t s = L_.stream;
    // algorithmic bytes of the encode: read 2^L, write 2^n words per column (16N(1 + 2^-B) for 4 columns),
    // split evenly over the passes
    const double enc_bytes = 4.0 * ncols * ((double)N + (double)((size_t)1 << L));
    if (L == 0) {
        Scope scope(L_, "ntt_broadcast", enc_bytes);
        dim3 grid((unsigned)((N + 255) / 256), ncols, L_.batch);
        ntt_broadcast_kernel<<<grid, 256, 0, s>>>(d_coef, coef_stride, d_out, out_stride, N, L_.bstride / 4);
        return 0;
    }
    // columns per workgroup: the largest divisor of ncols that is <= 4 (the 4 coordinate columns share every twiddle)
    const uint32_t cpw_max = L_.tune->ntt_cpw <= MAX_COLS_PER_WG ? L_.tune->ntt_cpw : MAX_COLS_PER_WG;  // tuning knob (1, 2 or 4)
    uint32_t cpw = cpw_max;
    while (ncols % cpw) cpw--;
    const size_t lds_bytes = (size_t)cpw * TILE_WORDS * sizeof(uint32_t);
    // (4 column tiles = 68 KiB of dynamic LDS, above the 64 KiB default: every context opts its device's function object in when it
    // is created, ntt_opt_in_dynamic_lds below)

    // real layers i = L-1 .. 0; the last pass takes up to 12 of them, the strided passes before it up to 8 each
    const uint32_t last_t = L < TILE_LOG ? L : TILE_LOG;
    uint32_t rest = L - last_t;
    const uint32_t mid_max = TILE_LOG - MID_LOG_W;
    const uint32_t n_mid_generic = (rest + mid_max - 1) / mid_max;
    uint32_t n_mid_fast = 0;  // strided passes of the fast plan (0: the generic plan's n_mid_generic)
    NttArgs a{};
    a.in = d_coef;
    a.in_stride = coef_stride;
    a.in_mask = (uint32_t)(((size_t)1 << L) - 1);
    a.out = d_out;
    a.out_stride = out_stride;
    a.tw = d_tw;
    a.n = n;
    a.init_y = ds.init_y;
    a.bstride_w = L_.bstride / 4;
    uint32_t cpw4 = MAX_COLS_PER_WG;
    // A workgroup takes its columns one after the other (they share the twiddle registers), so a pass is at least columns x one tile's
    // load - stages - store chain long: ~17 us for 4 columns however small the transform (2^18 and 2^20 domains take the same time).
    // While the launch cannot fill the chip anyway (< 512 tiles), one column per workgroup: four times the workgroups, a quarter of
    // the chain, the twiddles fetched four times (they are a few hundred KB at these sizes).
    const uint32_t cpw_small = L_.tune->ntt_cpw_small;  // tuning knob: columns per workgroup of launches below 512 tiles
    if ((N >> TILE_LOG) * (size_t)L_.batch < 512) cpw4 = cpw_small;  // (at 512 tiles the 4-column form is ahead again: 29 vs 32 us)
    while (ncols % cpw4) cpw4--;
    // one pass over layers a.i_hi .. a.i_lo
    uint32_t fused_levels = 0;
    auto launch_pass = [&](uint32_t t, const char* name) {
        const bool aligned = ((a.in_stride | a.out_stride) & 3) == 0 && (a.in_mask & 3u) == 3u &&
                             ((reinterpret_cast<uintptr_t>(a.in) | reinterpret_cast<uintptr_t>(a.out) | L_.bstride) & 15) == 0;
        if (sink && aligned && a.log_w == 0 && t == 12 && ncols == 4 && n >= TILE_LOG && out_log == n) {
            // the contiguous last pass of 12 layers over the 4 coordinate columns: fused with leaf hashing + 6 tree levels
            NttTreeArgs ta{};
            ta.a = a;
            ta.a.ncols = 4;
            ta.layers = sink->layers;
            ta.last_out = sink->last_out;
            ta.bstride = L_.bstride;
            // algorithmic bytes: this pass's share of the encode + leaves (16 B in, 32 B out) + 6 node levels (96 B per node)
            const bool reg_only = L_.tune->ntt_tree_reg_only;  // A/B knob
            const uint32_t levels = reg_only ? 5u : ENCODE_TREE_LEVELS;
            double bytes = enc_bytes / ((n_mid_fast ? n_mid_fast : n_mid_generic) + 1) + 48.0 * (double)N;
            for (uint32_t l = 1; l < levels; l++) bytes += 96.0 * (double)(N >> l);
            Scope scope(L_, reg_only ? "ntt_last_tree5" : "ntt_last_tree7", bytes);
            const dim3 grid((unsigned)(N >> TILE_LOG), 1, L_.batch);
            const bool tp = (size_t)grid.x * grid.z >= L_.tune->tp_min_wgs;  // (tree.hip tp_launch: the throughput form pays from ~3 waves per SIMD on)
#define FR_LAUNCH_LAST_TREE(SA, RO)                                                      \
    do {                                                                                 \
        if (tp)                                                                          \
            ntt_last_tree_kernel<SA, RO, true><<<grid, NTT_THREADS, 0, s>>>(ta);         \
        else                                                                             \
            ntt_last_tree_kernel<SA, RO, false><<<grid, NTT_THREADS, 0, s>>>(ta);        \
    } while (0)
            if (sink->layers && reg_only)
                FR_LAUNCH_LAST_TREE(true, true);
            else if (sink->layers)
                FR_LAUNCH_LAST_TREE(true, false);
            else if (reg_only)
                FR_LAUNCH_LAST_TREE(false, true);
            else
                FR_LAUNCH_LAST_TREE(false, false);
#undef FR_LAUNCH_LAST_TREE
            fused_levels = levels;
            return;
        }
        if (fsink && aligned && a.log_w == 0 && t == 12 && ncols == 4 && n >= TILE_LOG + 2 && out_log == n && L_.batch == 1 &&
            ((reinterpret_cast<uintptr_t>(fsink->line1) | reinterpret_cast<uintptr_t>(fsink->line2) | reinterpret_cast<uintptr_t>(fsink->itw)) & 15) == 0) {
            // the contiguous last pass over the 4 coordinate columns, folded twice in registers
            NttFoldArgs fa{};
            fa.a = a;
            fa.a.ncols = 4;
            fa.itw = fsink->itw;
            fa.inv_init_y = ds.inv_init_y;
            for (int i = 0; i < 4; i++) fa.alpha0[i] = fsink->alpha0[i], fa.alpha1[i] = fsink->alpha1[i];
            fa.line1 = fsink->line1;
            fa.line2 = fsink->line2;
            // algorithmic bytes: this pass's share of the encode + fold_circle_into_line (24 N) + fold_line at N / 2 (12 N)
            Scope scope(L_, "ntt_last_fold2", enc_bytes / ((n_mid_fast ? n_mid_fast : n_mid_generic) + 1) + 36.0 * (double)N);
            const dim3 grid((unsigned)(N >> TILE_LOG), 1, 1);
            // below 512 tiles the four columns run side by side in 1024-thread workgroups (the launch is a latency chain there);
            // needs the 68 KB dynamic-LDS opt-in the context obtained at creation (Tuning::lds_opt_in_ok)
            const bool side_by_side = (N >> TILE_LOG) < 512 && L_.tune->lds_opt_in_ok && L_.tune->ntt_cpw == 4 && !L_.tune->ntt_no_cp;
            const size_t cp_lds = (size_t)4 * TILE_WORDS * sizeof(uint32_t);
            if (side_by_side && fsink->accumulate)
                ntt_last_fold_cp_kernel<true><<<grid, NTT_CP_THREADS, cp_lds, s>>>(fa);
            else if (side_by_side)
                ntt_last_fold_cp_kernel<false><<<grid, NTT_CP_THREADS, cp_lds, s>>>(fa);
            else if (fsink->accumulate)
                ntt_last_fold_kernel<true><<<grid, NTT_THREADS, 0, s>>>(fa);
            else
                ntt_last_fold_kernel<false><<<grid, NTT_THREADS, 0, s>>>(fa);
            fused_levels = 1;
            return;
        }
        Scope scope(L_, name, enc_bytes / ((n_mid_fast ? n_mid_fast : n_mid_generic) + 1));
        // ntt_tile12_rep_kernel: one workgroup per source tile looping over the high blocks.  It cuts the pass's fetches from the memory
        // side to the unique coefficients; the pass is bound by its butterflies and LDS round trips, so that bought nothing in round 3
        // (142 vs 140 us at 2^24, profiles/r03_ntt_mid_rep_ab.txt) and costs a launch that no longer fills the chip its width
        // (a lone 2^22 blob: 256 workgroups, 33 vs 26 us).  With the round-5 kernels it is ahead where it still fills the chip:
        // 102 vs 106 us at 2^24 and the pass AFTER it 144 vs 150 (profiles/r05_prio_product_ab.txt, last block); streams of 2^22 / 2^20
        // blobs equal.  FRIEDA_NTT_REP: 0 = never, 1 = wherever the shape allows, 2 (default) = from 1024 workgroups on.
        const uint32_t hb_log = n - 1 - a.i_hi;  // high blocks = 2^hb_log
        const uint64_t rep_wgs = (uint64_t)((N >> TILE_LOG) >> (hb_log < 3 ? hb_log : 3)) * (ncols / 2) * L_.batch;
        const bool use_rep = L_.tune->ntt_rep == 1 || (L_.tune->ntt_rep == 2 && rep_wgs >= 1024);
        if (use_rep && out_log == n && aligned && t == 8 && a.log_w == MID_LOG_W && hb_log >= 1 && ((uint64_t)a.in_mask >> (a.i_hi + 1)) == 0 && ncols % 2 == 0) {
            // this pass reads the (replicated) coefficient vector: every high block has the same source tile
            a.rep_log = hb_log < 3 ? hb_log : 3;
            a.ncols = 2;
            dim3 grid((unsigned)((N >> TILE_LOG) >> a.rep_log), ncols / 2, L_.batch);
            ntt_tile12_rep_kernel<2><<<grid, NTT_THREADS, 0, s>>>(a);
            a.rep_log = 0;
        } else if (aligned && t + a.log_w == TILE_LOG && (t == 12 || t == 8 || t == 4)) {
            a.ncols = cpw4;
            dim3 grid((unsigned)(N >> TILE_LOG), ncols / cpw4, L_.batch);
            if (t == 12)
                ntt_tile12_kernel<3, 0><<<grid, NTT_THREADS, 0, s>>>(a);
            else if (t == 8)
                ntt_tile12_kernel<2, MID_LOG_W><<<grid, NTT_THREADS, 0, s>>>(a);
            else
                ntt_tile12_kernel<1, 8><<<grid, NTT_THREADS, 0, s>>>(a);  // one radix-16 stage, 1 KiB runs
        } else {
            a.ncols = cpw;
            set_stages(a, t);
            dim3 grid((unsigned)(N >> (t + a.log_w)), ncols / cpw, L_.batch);
            ntt_tile_kernel<<<grid, NTT_THREADS, lds_bytes, s>>>(a);
        }
    };
    a.in_limit = a.in_mask + 1;
    uint32_t i_hi = L - 1;
    // The strided layers run as passes of the fast kernel: 8 layers (two radix-16 stages, 64-byte runs) or 4 layers (one stage, 1 KiB
    // runs), a 4-layer pass first when the number of 4-layer units is odd.  `rest` is rounded up to a multiple of 4 with up to three of the
    // zero-padded layers above the coefficient vector: the first pass executes them as real butterflies against zero coefficients (source
    // words beyond 2^L read as zero instead of wrapping), which costs less than the generic kernel saves (44.8 -> 28 us at 2^22, 623 ->
    // 3xx us for the 9 strided layers of a 2^25 domain).  Needs padz <= log_blowup_factor and 16-byte aligned buffers; otherwise the
    // generic passes below take over.
    const bool no_pad8 = L_.tune->ntt_no_pad8;  // A/B knob: generic kernel unless a pass has exactly 8 real layers
    const uint32_t units = (rest + 3) / 4, padz = 4 * units - rest;
    const bool base_aligned = ((a.in_stride | a.out_stride) & 3) == 0 &&
                              ((reinterpret_cast<uintptr_t>(a.in) | reinterpret_cast<uintptr_t>(a.out) | L_.bstride) & 15) == 0;
    if (rest > 0 && base_aligned && padz <= out_log - L && L >= 4 && (!no_pad8 || (padz == 0 && (units & 1) == 0))) {
        n_mid_fast = (units + 1) / 2;
        uint32_t top = L - 1 + padz, left = units;
        bool first = true;
        while (left) {
            const uint32_t t = (left & 1) ? 4u : 8u;
            a.i_hi = top;
            a.i_lo = top + 1 - t;
            a.log_w = TILE_LOG - t;
            if (first) {
                a.in_mask = (uint32_t)(((size_t)1 << (L + padz)) - 1);
                a.in_limit = (uint32_t)1 << L;
            }
            launch_pass(t, "ntt_pass_mid");
            a.in = d_out;
            a.in_stride = out_stride;
            a.in_mask = (uint32_t)(N - 1);
            a.in_limit = (uint32_t)N;
            top -= t;
            left -= t / 4;
            first = false;
        }
        rest = 0;
        i_hi = last_t - 1;
    }
    const uint32_t n_mid = n_mid_generic;
    for (uint32_t p = 0; p < n_mid && rest; p++) {
        uint32_t t = (rest + (n_mid - p) - 1) / (n_mid - p);
        a.i_hi = i_hi;
        a.i_lo = i_hi + 1 - t;
        // runs of 2^log_w consecutive words: 64 bytes for a full pass of 8 layers; a pass of fewer layers takes longer runs so
        // that its tile still has 4096 words (i_lo >= last_t = 12 >= log_w whenever a strided pass exists)
        a.log_w = t >= mid_max ? MID_LOG_W : TILE_LOG - t;
        launch_pass(t, "ntt_pass_mid");
        a.in = d_out;
        a.in_stride = out_stride;
        a.in_mask = (uint32_t)(N - 1);
        a.in_limit = (uint32_t)N;
        rest -= t;
        i_hi = a.i_lo - 1;
    }
    a.i_hi = i_hi;  // == last_t - 1
    a.i_lo = 0;
    a.log_w = 0;
    launch_pass(last_t, "ntt_pass_last");
    return fused_levels;
}
}  // namespace

}  // namespace k
}  // namespace frieda
