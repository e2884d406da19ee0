// tree.hip — the fused commit-phase kernels: multi-level Merkle subtrees with the FRI fold folded into leaf hashing,
// the on-device Fiat–Shamir step, the single-workgroup FRI tail and the device-transcript grind (gfx950).
//
// Reference path: `MerkleProver::commit` over the 4 coordinate columns of each FRI layer, `FriOps::fold_*`,
// `MerkleChannel::mix_root` + `Channel::draw_felt`, `FriProver::commit_last_layer`, `GrindOps::grind`
// (/root/reference/src/commit.rs:17-21, src/proof.rs:52-59; stwo core/fri.rs, core/vcs/prover.rs).
//
// tree5r_kernel  (launches of >= 2^20 level-A nodes) one workgroup owns 1024 consecutive level-A nodes and produces five tree
//                levels (1024, 512, 256, 128, 64 nodes).  A thread owns four adjacent level-A nodes and hashes them, their two
//                parents and their grandparent in registers; only the last two levels cross threads (LDS, struct-of-arrays:
//                one 4-byte column per hash word, so ds_write_b32 / ds_read_b64 are bank-conflict free).
// tree5_kernel   (smaller launches) 256 level-A nodes per workgroup, one per thread, all levels through LDS: more, smaller
//                workgroups and a shorter dependent chain when the launch cannot fill the chip.
//                Level A is produced, by mode, from
//                  LEAF4        the 4 SoA columns of a layer (leaf = H(c0,c1,c2,c3, 0 x 12)),
//                  NODE         the hashes of the level below (node = H(left || right)),
//                  FOLD_CIRCLE  / FOLD_LINE  the previous FRI layer: the fold of pair (2g, 2g+1) is computed in registers,
//                               written to the new layer's columns and hashed at once — the folded layer is never re-read.
//                Every level is stored (generate_proof needs the layers for decommitment) or only the last (commit).
// tree7q_kernel  the narrow middle of a tree (level-A sizes <= 2^15): 64 nodes per workgroup, up to seven levels, every node
//                hashed by a quad of lanes (4-lane cooperative Blake2s) — ~1.5 us per level, wide levels spread over many CUs.
// top_kernel     one workgroup finishes a tree from <= 512 hashes to the root with quad hashing and then, in quad 0, mixes
//                the root into the device transcript and draws the next folding alpha.
// tail_kernel    one workgroup runs every remaining FRI layer of <= 2048 points: fold, tree, channel, ... then interpolates
//                the last layer (line iFFT in LDS), enforces stwo's degree assertion and mixes the polynomial.
// grind_dev      proof-of-work scan keyed by the digest in the device transcript.
//
// Roofline: Blake2s compression is ~960 integer VALU instructions of which half (v_alignbit_b32, v_add3_u32) issue at
// half rate on gfx950 (profiles/r01_valu_rate_mi355x.txt): ~1456 full-rate slots per 64 B hashed, i.e. the kernels are
// bound by the integer VALU pipe (~38 G compressions/s chip-wide), not by HBM.
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cstdlib>

#include "clock_stamps.h"
#include "blake2s.h"
#include "dev_transcript.h"
#include "kernels.h"
#include "tree_dev.h"

namespace frieda {
namespace k {

namespace {

// The latency-bound kernels (a chain of dependent compressions on a handful of waves) often share the chip with the
// chip-filling kernels of another proof in flight.  A raised wave priority lets their few waves win the SIMD's issue arbitration
// (priority, then age) against the co-resident throughput waves, which lose next to nothing: the chain is <= 8 waves per CU.
__device__ __forceinline__ void latency_kernel_priority() {
#ifndef FRIEDA_NO_LATENCY_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
}

constexpr int T5_THREADS = 256;
constexpr uint32_t T5_UNITS = 1024;
constexpr uint32_t T5_LEVELS = 5;
constexpr int WG1_THREADS = 512;  // single-workgroup kernels: 2 waves per SIMD saturate the VALU; 256 VGPRs per lane for the quad hash

using namespace treedev;

__device__ __forceinline__ uint32_t inv_circle_twiddle(const uint32_t* __restrict__ itw, uint32_t n, size_t i, uint32_t inv_init_y) {
    if (n < 3) return (i & 1u) ? m31_neg(inv_init_y) : inv_init_y;
    size_t j = i >> 2;
    uint32_t r = (uint32_t)(i & 3u);
    uint32_t v = itw[2 * j + (r < 2 ? 1 : 0)];
    return (r == 1 || r == 2) ? m31_neg(v) : v;
}

// fold of the adjacent pair (2g, 2g+1) of a 4-column SoA layer: f0 + alpha * f1 with (f0, f1) = (a + b, (a - b) * itw)
__device__ __forceinline__ QM31 fold_pair(const uint32_t* __restrict__ src, size_t stride, size_t g, uint32_t it, const QM31Mat& alpha_m) {
    uint2 a = reinterpret_cast<const uint2*>(src)[g];
    uint2 b = reinterpret_cast<const uint2*>(src + stride)[g];
    uint2 c = reinterpret_cast<const uint2*>(src + 2 * stride)[g];
    uint2 d = reinterpret_cast<const uint2*>(src + 3 * stride)[g];
    QM31 x = {a.x, b.x, c.x, d.x}, y = {a.y, b.y, c.y, d.y};
    return qm_fold_pair(x, y, it, alpha_m);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// tree5
// ------------------------------------------------------------------------------------------------
enum TreeMode { T_LEAF4 = 0, T_NODE = 1, T_FOLD_CIRCLE = 2, T_FOLD_LINE = 3 };

struct TreeArgs {
    const uint32_t* cols;    // LEAF4: this layer's columns; FOLD: the previous layer's columns
    size_t col_stride;
    const uint8_t* children;  // NODE: hashes of the level below A (two per A node)
    uint32_t* out_vals;       // FOLD: the new layer's columns
    size_t out_stride;
    const uint32_t* itw;  // FOLD_CIRCLE: inverse table base; FOLD_LINE: the inverse level of the previous layer's domain
    uint32_t n;           // circle domain log size
    uint32_t inv_init_y;
    const DevTranscript* tr;  // FOLD: alpha
    uint32_t level_a;         // layer index of level A (2^level_a nodes in the launch)
    uint32_t tree_log;        // log size of the tree (leaves-first offsets)
    uint8_t* layers;          // store_all: tree storage base
    uint8_t* last_out;        // !store_all: destination of the last produced level (indexed by global node index)
    int store_all;
    int skip_a;               // store_all, but level A itself is not written (nothing reads the leaf hashes of a FRI layer's tree:
                              // a decommitment opens both members of every queried pair, so the verifier derives them from values)
    int skip_bc;              // store_all + skip_a, large trees (Tuning::tree_skip_log): the register-subtree kernel does not write the two
                              // levels above the leaves either — 3/4 of a tree's bytes, read back only along ~20 opened paths, which
                              // decommit.hip re-hashes from the layer's values instead (2.7 % of a 2^24 proof, profiles/r05_skip_levels.txt)
    size_t bstride;           // batch: bytes between consecutive blobs' workspaces (blob = blockIdx.y); tr is an array
};

// the arguments of the blob this workgroup works on
__device__ __forceinline__ void tree_args_of_blob(TreeArgs& a) {
    const size_t off = (size_t)blockIdx.y * a.bstride;
    a.cols = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.cols) + off);
    a.children = a.children + off;
    a.out_vals = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(a.out_vals) + off);
    a.layers = a.layers + off;
    a.last_out = a.last_out + off;
    a.tr = a.tr + blockIdx.y;
}

namespace {

// TP: the compressions in the throughput form (blake2s.h) — chosen by the launcher for launches that fill the chip (tp_launch): below
// ~3 waves per SIMD the run structure buys nothing and costs latency (a lone 2^20 proof 0.555 -> 0.600 ms with the throughput form everywhere)
// In the chip-filling launches a wave runs its prologue — the loads of its inputs and the fold arithmetic — at a priority above the
// compressions' slow runs (blake2s.h): its loads are in flight before it competes for the vector pipe, and the folds do not queue
// behind eight waves' worth of compressions (tree5_fold_circle 335 -> 296 us, tree5_fold_line 553 -> 520, tree5_leaf 528 -> 512;
// profiles/r05_prio_product_ab.txt).  The first compression's first statement sets the compressions' own priorities.
#ifndef FRIEDA_T5R_LOAD_PRIO
#define FRIEDA_T5R_LOAD_PRIO 3
#endif
__device__ __forceinline__ void tree_prologue_priority() {
    if constexpr (FRIEDA_T5R_LOAD_PRIO != 0) __builtin_amdgcn_s_setprio(FRIEDA_T5R_LOAD_PRIO);
}

template <int MODE, uint32_t UNITS, bool TP>
__global__ __launch_bounds__(T5_THREADS) void tree5_kernel(TreeArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t RA[8 * (UNITS + 4)];
    __shared__ __attribute__((aligned(16))) uint32_t RB[8 * (UNITS / 2 + 4)];
    tree_args_of_blob(a);
    const uint32_t t = threadIdx.x;
    const size_t total_a = (size_t)1 << a.level_a;
    const size_t wg_base = (size_t)blockIdx.x * UNITS;
    const uint32_t cnt_a = (uint32_t)(total_a - wg_base < UNITS ? total_a - wg_base : UNITS);  // a power of two
    uint32_t nl = 1;
    while (nl < T5_LEVELS && (cnt_a >> nl) >= 1) nl++;

    QM31Mat alpha = {};  // multiplication by the layer's folding challenge (uniform: lives in scalar registers)
    if (MODE == T_FOLD_CIRCLE || MODE == T_FOLD_LINE) alpha = qm_matrix({a.tr->alpha[0], a.tr->alpha[1], a.tr->alpha[2], a.tr->alpha[3]});

    // ---- level A ----
    {
        const bool last = nl == 1;
        uint8_t* gout = a.store_all ? (a.skip_a ? nullptr : a.layers + layer_off(a.tree_log, a.level_a)) : (last ? a.last_out : nullptr);
        for (uint32_t j = t; j < cnt_a; j += T5_THREADS) {
            const size_t g = wg_base + j;
            uint32_t h[8];
            if (MODE == T_NODE) {
                uint32_t m[16];
                load_children(a.children, g, m);
                b2_merkle_block<TP ? FRIEDA_B2_IDLE_NODE : B2_LAT>(m, h);
            } else if (MODE == T_LEAF4) {
                leaf_hash<TP ? FRIEDA_B2_IDLE_LEAF : B2_LAT>(a.cols[g], a.cols[a.col_stride + g], a.cols[2 * a.col_stride + g], a.cols[3 * a.col_stride + g], h);
            } else {
                uint32_t it = (MODE == T_FOLD_CIRCLE) ? inv_circle_twiddle(a.itw, a.n, g, a.inv_init_y) : a.itw[g];
                QM31 r = fold_pair(a.cols, a.col_stride, g, it, alpha);
                a.out_vals[g] = r.a;
                a.out_vals[a.out_stride + g] = r.b;
                a.out_vals[2 * a.out_stride + g] = r.c;
                a.out_vals[3 * a.out_stride + g] = r.d;
                leaf_hash<TP ? FRIEDA_B2_IDLE_LEAF : B2_LAT>(r.a, r.b, r.c, r.d, h);
            }
            if (gout) store_hash(gout, g, h);
            if (!last) lds_put(RA, cnt_a + 4, j, h);
        }
        __syncthreads();
    }
    // ---- levels B..E from LDS ----
    for (uint32_t l = 1; l < nl; l++) {
        const uint32_t cnt = cnt_a >> l;
        uint32_t* dst = (l & 1) ? RB : RA;
        const uint32_t* src = (l & 1) ? RA : RB;
        const bool last = l + 1 == nl;
        uint8_t* gout = a.store_all ? a.layers + layer_off(a.tree_log, a.level_a - l) : (last ? a.last_out : nullptr);
        for (uint32_t j = t; j < cnt; j += T5_THREADS) {
            uint32_t m[16], h[8];
            lds_children(src, 2 * cnt + 4, j, m);
            b2_merkle_block<TP ? FRIEDA_B2_IDLE_NODE : B2_LAT>(m, h);
            if (gout) store_hash(gout, (wg_base >> l) + j, h);
            if (!last) lds_put(dst, cnt + 4, j, h);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// tree5r: the same five levels for the large launches, with the three lowest levels in registers
// ------------------------------------------------------------------------------------------------
// A thread owns FOUR ADJACENT level-A nodes (4t .. 4t+3 of the workgroup's 1024): it produces them, their two parents and
// their grandparent — seven compressions — without LDS or a barrier; inputs are one 16-byte load per column (or two for a
// fold).  Only the 256 grandparents of the workgroup go through LDS for the last two levels (128 and 64 nodes), so a workgroup
// needs 12 KB instead of 48 KB of LDS and the CU holds eight of them: while one is in its narrow levels the others fill the VALU.
// REG_ONLY: the three register levels alone (1024, 512, 256 nodes per workgroup; no LDS, no barrier).  The workgroups of a launch
// start together and stay in step, so the two LDS levels — half, then a quarter of the waves busy — are phases in which the
// whole chip runs at 50 % / 25 %; leaving them to the next (eight times smaller) launch keeps the big launch at full rate.
FR_CLOCK_DECL(g_clock_tree5r)
template <int MODE, bool REG_ONLY, bool TP>
__global__ __launch_bounds__(T5_THREADS) void tree5r_kernel(TreeArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t RC[8 * (256 + 4)];
    __shared__ __attribute__((aligned(16))) uint32_t RD[8 * (128 + 4)];
    tree_args_of_blob(a);
    uint32_t t_tie = threadIdx.x;
    FR_CLOCK_BEGIN(t_tie)
    const uint32_t t = t_tie;
    const size_t wg_base = (size_t)blockIdx.x * 1024;  // the launcher guarantees 2^level_a >= 1024
    const size_t g0 = wg_base + 4 * t;
    uint8_t* out_a = a.store_all && !a.skip_a ? a.layers + layer_off(a.tree_log, a.level_a) : nullptr;
    // (skip_bc: levels B and C stay unwritten; the register-only variant hands level C to the next launch and keeps it)
    uint8_t* out_b = a.store_all && !a.skip_bc ? a.layers + layer_off(a.tree_log, a.level_a - 1) : nullptr;
    uint8_t* out_c = a.store_all ? (a.skip_bc && !REG_ONLY ? nullptr : a.layers + layer_off(a.tree_log, a.level_a - 2)) : (REG_ONLY ? a.last_out : nullptr);
    uint8_t* out_d = a.store_all ? a.layers + layer_off(a.tree_log, a.level_a - 3) : nullptr;
    uint8_t* out_e = a.store_all ? a.layers + layer_off(a.tree_log, a.level_a - 4) : a.last_out;

    // the four level-A inputs of this thread, one uint4 per column: loaded (LEAF4) or folded from the previous layer (FOLD)
    if (TP) tree_prologue_priority();
    uint4 lc0 = {}, lc1 = {}, lc2 = {}, lc3 = {};
    if (MODE == T_LEAF4) {
        lc0 = *reinterpret_cast<const uint4*>(a.cols + g0);
        lc1 = *reinterpret_cast<const uint4*>(a.cols + a.col_stride + g0);
        lc2 = *reinterpret_cast<const uint4*>(a.cols + 2 * a.col_stride + g0);
        lc3 = *reinterpret_cast<const uint4*>(a.cols + 3 * a.col_stride + g0);
    } else if (MODE == T_FOLD_CIRCLE || MODE == T_FOLD_LINE) {
        const QM31Mat alpha = qm_matrix({a.tr->alpha[0], a.tr->alpha[1], a.tr->alpha[2], a.tr->alpha[3]});
        uint32_t it[4];
        if (MODE == T_FOLD_CIRCLE) {
            // inverse circle twiddles of pairs 4j .. 4j+3 are [y, -y, -x, x] of the table pair j (n >= 3 here)
            const uint2 xy = *reinterpret_cast<const uint2*>(a.itw + 2 * (g0 >> 2));
            it[0] = xy.y, it[1] = m31_neg(xy.y), it[2] = m31_neg(xy.x), it[3] = xy.x;
        } else {
            const uint4 w = *reinterpret_cast<const uint4*>(a.itw + g0);
            it[0] = w.x, it[1] = w.y, it[2] = w.z, it[3] = w.w;
        }
        // source values 2 g0 .. 2 g0 + 7 of every column (32 contiguous bytes per lane): the pairs (2g, 2g+1), g = g0 .. g0+3
        uint4 s[4][2];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            s[c][0] = *reinterpret_cast<const uint4*>(a.cols + c * a.col_stride + 2 * g0);
            s[c][1] = *reinterpret_cast<const uint4*>(a.cols + c * a.col_stride + 2 * g0 + 4);
        }
        const QM31 r0 = qm_fold_pair({s[0][0].x, s[1][0].x, s[2][0].x, s[3][0].x}, {s[0][0].y, s[1][0].y, s[2][0].y, s[3][0].y}, it[0], alpha);
        const QM31 r1 = qm_fold_pair({s[0][0].z, s[1][0].z, s[2][0].z, s[3][0].z}, {s[0][0].w, s[1][0].w, s[2][0].w, s[3][0].w}, it[1], alpha);
        const QM31 r2 = qm_fold_pair({s[0][1].x, s[1][1].x, s[2][1].x, s[3][1].x}, {s[0][1].y, s[1][1].y, s[2][1].y, s[3][1].y}, it[2], alpha);
        const QM31 r3 = qm_fold_pair({s[0][1].z, s[1][1].z, s[2][1].z, s[3][1].z}, {s[0][1].w, s[1][1].w, s[2][1].w, s[3][1].w}, it[3], alpha);
        lc0 = make_uint4(r0.a, r1.a, r2.a, r3.a);
        lc1 = make_uint4(r0.b, r1.b, r2.b, r3.b);
        lc2 = make_uint4(r0.c, r1.c, r2.c, r3.c);
        lc3 = make_uint4(r0.d, r1.d, r2.d, r3.d);
        *reinterpret_cast<uint4*>(a.out_vals + g0) = lc0;  // the folded layer, full 16-byte stores
        *reinterpret_cast<uint4*>(a.out_vals + a.out_stride + g0) = lc1;
        *reinterpret_cast<uint4*>(a.out_vals + 2 * a.out_stride + g0) = lc2;
        *reinterpret_cast<uint4*>(a.out_vals + 3 * a.out_stride + g0) = lc3;
    }
    uint32_t hb[2][8];
    // (the two halves are written out through a lambda: the unroller refuses loops whose body carries the compressions' inline asm)
    auto half_ab = [&](auto half_c) {
        constexpr int half = decltype(half_c)::value;
        uint32_t ha[2][8];
        const size_t g = g0 + 2 * half;  // level-A nodes g, g + 1
        if (MODE == T_NODE) {
            {
                uint32_t m[16];
                load_children(a.children, g, m);
                b2_merkle_block<TP ? FRIEDA_B2_IDLE_NODE : B2_LAT>(m, ha[0]);
            }
            {
                uint32_t m[16];
                load_children(a.children, g + 1, m);
                b2_merkle_block<TP ? FRIEDA_B2_IDLE_NODE : B2_LAT>(m, ha[1]);
            }
        } else if (half == 0) {
            leaf_hash<TP ? FRIEDA_B2_IDLE_LEAF : B2_LAT>(lc0.x, lc1.x, lc2.x, lc3.x, ha[0]);
            leaf_hash<TP ? FRIEDA_B2_IDLE_LEAF : B2_LAT>(lc0.y, lc1.y, lc2.y, lc3.y, ha[1]);
        } else {
            leaf_hash<TP ? FRIEDA_B2_IDLE_LEAF : B2_LAT>(lc0.z, lc1.z, lc2.z, lc3.z, ha[0]);
            leaf_hash<TP ? FRIEDA_B2_IDLE_LEAF : B2_LAT>(lc0.w, lc1.w, lc2.w, lc3.w, ha[1]);
        }
        if (out_a) {
            store_hash(out_a, g, ha[0]);
            store_hash(out_a, g + 1, ha[1]);
        }
        uint32_t m[16];
#pragma unroll
        for (int w = 0; w < 8; w++) m[w] = ha[0][w], m[8 + w] = ha[1][w];
        b2_merkle_block<TP ? FRIEDA_B2_IDLE_NODE : B2_LAT>(m, hb[half]);
    };
    half_ab(std::integral_constant<int, 0>{});
    half_ab(std::integral_constant<int, 1>{});
    if (out_b) {
        store_hash(out_b, (g0 >> 1), hb[0]);
        store_hash(out_b, (g0 >> 1) + 1, hb[1]);
    }
    uint32_t hc[8];
    {
        uint32_t m[16];
#pragma unroll
        for (int w = 0; w < 8; w++) m[w] = hb[0][w], m[8 + w] = hb[1][w];
        b2_merkle_block<TP ? FRIEDA_B2_IDLE_NODE : B2_LAT>(m, hc);
    }
    if (out_c) store_hash(out_c, g0 >> 2, hc);
    if (REG_ONLY) return;
    lds_put(RC, 256 + 4, t, hc);
    __syncthreads();
    if (t < 128) {
        uint32_t m[16], h[8];
        lds_children(RC, 256 + 4, t, m);
        b2_merkle_block<TP ? FRIEDA_B2_IDLE_NODE : B2_LAT>(m, h);
        if (out_d) store_hash(out_d, (wg_base >> 3) + t, h);
        lds_put(RD, 128 + 4, t, h);
    }
    __syncthreads();
    if (t < 64) {
        uint32_t m[16], h[8];
        lds_children(RD, 128 + 4, t, m);
        b2_merkle_block<TP ? FRIEDA_B2_IDLE_NODE : B2_LAT>(m, h);
        store_hash(out_e, (wg_base >> 4) + t, h);
        FR_CLOCK_END(g_clock_tree5r, h[0])
    }
}

// ------------------------------------------------------------------------------------------------
// four-lane Blake2s compression for the latency-bound top of a tree
// ------------------------------------------------------------------------------------------------
// A lone wave issues one VALU instruction every ~4 cycles whatever its rate class, so one compression per lane costs ~1.8 us
// (~960 instructions at 2.4 GHz) however few hashes a level has.  For levels of <= 128 nodes the four lanes of an aligned quad
// share one compression instead (~0.95 us, tools/quad_latency.hip): lane q owns state column q (v[q], v[4+q], v[8+q],
// v[12+q]); the column step is lane-local, the diagonal step rotates b, c, d by 1, 2, 3 lanes with DPP quad_perm moves.
// Messages live in LDS in the quad layout: hash j at words [QS j, QS j + 8), so the two children of node j form the 18-word
// message slot [2 QS j, 2 QS j + 18) with message word idx at idx + (idx >> 3) (9 words per hash also spreads the quads over
// the banks).  Two such buffers alternate level by level (children in buffer P, parents into buffer P ^ 1 at hash position j,
// i.e. straight into the parent's message slot).  A quad keeps its node index through all levels, so every lane holds the forty
// LDS addresses of its message words (round r, fetch k) in registers for the whole kernel and the buffer parity is a compile-time
// immediate of the ds_read; the Blake2s IV words of its column are registers too (nothing is re-read from memory per level).
// Lane q ends up with output words q and 4 + q.
constexpr uint32_t QS = 9;                      // words between consecutive hashes in the quad (array-of-structs) LDS layout
constexpr uint32_t QBUF_WORDS = 256 * QS + 12;  // one buffer: 256 hashes = 128 message slots (a multiple of four words)

template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
constexpr int QROT1 = 0x39;  // lane q reads lane q+1
constexpr int QROT2 = 0x4E;  // lane q reads lane q+2
constexpr int QROT3 = 0x93;  // lane q reads lane q+3

// byte offsets inside a message slot of the message word that lane q = 0..3 needs at (round r, fetch k), one byte per lane;
// k = 0, 1: column step (SIGMA[r][2q + k]); k = 2, 3: diagonal step (SIGMA[r][8 + 2q + (k & 1)])
constexpr uint32_t sigma_pack(int r, int k) {
    uint32_t c = 0;
    for (int q = 0; q < 4; q++) {
        const uint32_t idx = b2detail::SIGMA[r][(k < 2 ? 0 : 8) + 2 * q + (k & 1)];
        c |= (4u * (idx + (idx >> 3))) << (8 * q);
    }
    return c;
}
__device__ __forceinline__ uint32_t sel4(uint32_t q, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
    uint32_t v = v0;
    v = q == 1 ? v1 : v;
    v = q == 2 ? v2 : v;
    v = q == 3 ? v3 : v;
    return v;
}

struct QuadCtx {
    uint32_t a[40];   // byte offset, inside a buffer, of this lane's message word for (round r, fetch k) at a[4 r + k]
    uint32_t wr;      // byte offset, inside a buffer, of word q of hash `quad` (the quad's output position)
    uint32_t c0, d0;  // IV[q], IV[4 + q]
};
// `quad`: the node index this quad keeps through the levels (its message slot); q = lane & 3
__device__ __forceinline__ void quad_ctx_init(QuadCtx& x, uint32_t quad, uint32_t q) {
    const uint32_t slot = 4u * 2u * QS * quad;
#pragma unroll
    for (int r = 0; r < 10; r++) {
#pragma unroll
        for (int k = 0; k < 4; k++) x.a[4 * r + k] = slot + ((sigma_pack(r, k) >> (8u * q)) & 0xFFu);
    }
    x.wr = 4u * (QS * quad + q);
    x.c0 = sel4(q, b2detail::IV[0], b2detail::IV[1], b2detail::IV[2], b2detail::IV[3]);
    x.d0 = sel4(q, b2detail::IV[4], b2detail::IV[5], b2detail::IV[6], b2detail::IV[7]);
}

struct Quad2 {
    uint32_t lo, hi;  // output words q and 4 + q
};

// message = the quad's slot in buffer P; state in: a = h[q], b = h[4+q], c = IV[q], d = IV[4+q] ^ {t0, t1, f0, f1}[q]
template <int P>
__device__ __forceinline__ Quad2 b2_compress_quad(const uint32_t* QQ, const QuadCtx& x, uint32_t ha, uint32_t hb, uint32_t c, uint32_t d) {
    uint32_t a = ha, b = hb;
    const char* mbase = reinterpret_cast<const char*>(QQ + P * QBUF_WORDS);
    auto fetch = [&](int i) { return *reinterpret_cast<const uint32_t*>(mbase + x.a[i]); };
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t m0 = fetch(4 * r), m1 = fetch(4 * r + 1), m2 = fetch(4 * r + 2), m3 = fetch(4 * r + 3);
        FR_B2_G(a, b, c, d, m0, m1);
        b = quad_perm<QROT1>(b);
        c = quad_perm<QROT2>(c);
        d = quad_perm<QROT3>(d);
        FR_B2_G(a, b, c, d, m2, m3);
        b = quad_perm<QROT3>(b);
        c = quad_perm<QROT2>(c);
        d = quad_perm<QROT1>(d);
    }
    return {ha ^ a ^ c, hb ^ b ^ d};
}

__device__ __forceinline__ void q_put_hash(uint32_t* Q, uint32_t j, const uint32_t (&h)[8]) {
#pragma unroll
    for (int w = 0; w < 8; w++) Q[QS * j + w] = h[w];
}
// `count` hashes (32-byte array of structs in global memory) -> quad layout, 16 bytes per lane and step
__device__ __forceinline__ void q_load_hashes(uint32_t* Q, const uint8_t* in, uint32_t count, uint32_t t, uint32_t nthreads) {
    const uint4* in4 = reinterpret_cast<const uint4*>(in);
    for (uint32_t e = t; e < 2 * count; e += nthreads) {
        const uint4 v = in4[e];
        uint32_t* dst = Q + QS * (e >> 1) + 4 * (e & 1);
        dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
    }
}
// store words q and 4+q of hash j (32-byte array-of-structs in global memory)
__device__ __forceinline__ void store_hash_quad(uint8_t* out, size_t j, uint32_t q, Quad2 v) {
    uint32_t* p = reinterpret_cast<uint32_t*>(out + 32 * j);
    p[q] = v.lo;
    p[4 + q] = v.hi;
}

// One quad level: the Merkle node (Blake2sMerkleHasher::hash_node: h = 0, t = f = 0) of this quad from its message slot in
// buffer P; the result goes to global memory (`gout` non-null, node index `node`) and, when `keep`, to hash position `quad`
// of buffer P ^ 1.  Ends with the workgroup barrier.
template <int P>
__device__ __forceinline__ void quad_level(uint32_t* QQ, const QuadCtx& x, bool active, uint8_t* gout, size_t node, uint32_t q, bool keep) {
    if (active) {
        const Quad2 v = b2_compress_quad<P>(QQ, x, 0u, 0u, x.c0, x.d0);
        if (gout) store_hash_quad(gout, node, q, v);
        if (keep) {
            uint32_t* w = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(QQ + (P ^ 1) * QBUF_WORDS) + x.wr);
            w[0] = v.lo;
            w[4] = v.hi;
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// single-workgroup tree machinery: LDS regions and the level loop
// ------------------------------------------------------------------------------------------------
// Levels of >= 512 hashes live in LDS as struct-of-arrays (S regions, one thread per parent node); levels of <= 256 hashes
// live in buffer 0 of the quad buffers QQ because their parents (<= 128 nodes = the quads of the workgroup) are quad-hashed.
//
// Reduces a level of 2^log_count hashes already in LDS (SoA in `s_cur` when 2^log_count >= 512, else quad layout in buffer 0
// of QQ) down to the root.  Every produced level l is stored at its leaves-first offset when `layers` is non-null.
// Returns the LDS address of the root (8 consecutive words), valid for all threads after the final barrier.
// `x` must have been initialised with quad = threadIdx.x >> 2.
__device__ const uint32_t* wg_reduce(uint32_t* s_cur, uint32_t* s_other, uint32_t* QQ, uint32_t log_count, uint8_t* layers,
                                     uint32_t tree_log, const QuadCtx& x) {
    const uint32_t t = threadIdx.x, q = t & 3, quad = t >> 2;
    int l = (int)log_count - 1;
    for (; l >= 8; l--) {
        // one node per thread; children in SoA
        const uint32_t cnt = 1u << l;
        uint8_t* gout = layers ? layers + layer_off(tree_log, (uint32_t)l) : nullptr;
        for (uint32_t j = t; j < cnt; j += WG1_THREADS) {
            uint32_t m[16], h[8];
            lds_children(s_cur, 2 * cnt + 4, j, m);
            b2_merkle_block<B2_LAT>(m, h);
            if (gout) store_hash(gout, j, h);
            if (cnt >= 512)
                lds_put(s_other, cnt + 4, j, h);
            else
                q_put_hash(QQ, j, h);  // 256 hashes: the parents are quad-hashed
        }
        uint32_t* tmp = s_cur;
        s_cur = s_other;
        s_other = tmp;
        __syncthreads();
    }
    // one node per quad (2^l <= 128 = WG1_THREADS / 4), buffers alternating
    uint32_t par = 0;
    while (l >= 0) {
        quad_level<0>(QQ, x, quad < (1u << l), layers ? layers + layer_off(tree_log, (uint32_t)l) : nullptr, quad, q, true);
        par = 1;
        if (--l < 0) break;
        quad_level<1>(QQ, x, quad < (1u << l), layers ? layers + layer_off(tree_log, (uint32_t)l) : nullptr, quad, q, true);
        par = 0;
        --l;
    }
    return QQ + par * QBUF_WORDS;
}

// Fiat–Shamir step after a root, by quad 0 (lanes 0..3 only): Blake2sMerkleChannel::mix_root (standard Blake2s-256 of
// digest || root) and Channel::draw_felt (digest || counter, retried until all eight words are < 2P).  The channel words this
// needs are read at kernel start (ChanRegs) so that no global load sits between the root and the alpha; quad 0's message slot
// in buffer 0 of QQ serves as the message buffer (the tree is finished by then).  Lane q returns alpha coordinate q.
struct ChanRegs {
    uint32_t dg_lo, dg_hi;  // digest words q and 4 + q
    uint32_t bound;         // draw_felt acceptance bound
    uint32_t n_roots;
    uint32_t n_chal;        // Channel::n_challenges
};
__device__ __forceinline__ void chan_prefetch(ChanRegs& cr, const DevTranscript* tr, uint32_t q) {
    cr.dg_lo = tr->ch.digest[q];
    cr.dg_hi = tr->ch.digest[4 + q];
    cr.bound = tr->draw_bound;
    cr.n_roots = tr->n_roots;
    cr.n_chal = tr->ch.n_challenges;
}
__device__ uint32_t channel_after_root_quad(DevTranscript* tr, ChanRegs& cr, uint32_t root_lo, uint32_t root_hi, uint32_t* QQ,
                                            const QuadCtx& x) {
    const uint32_t q = threadIdx.x & 3;
    // message = digest (words 0..7) || root (words 8..15) in the idx + (idx >> 3) layout
    QQ[q] = cr.dg_lo;
    QQ[4 + q] = cr.dg_hi;
    QQ[QS + q] = root_lo;
    QQ[QS + 4 + q] = root_hi;
    // h = IV ^ parameter block (digest 32, fanout 1, depth 1); t0 = 64 bytes; f0 = ~0 (single, final block)
    const uint32_t hq = x.c0 ^ (q == 0 ? 0x01010020u : 0u), h4q = x.d0;
    const uint32_t dflag = (q == 0) ? 64u : (q == 2 ? 0xFFFFFFFFu : 0u);
    const Quad2 dg = b2_compress_quad<0>(QQ, x, hq, h4q, x.c0, x.d0 ^ dflag);
    // draw_felt on the new digest
    uint32_t n_sent = 0;
    Quad2 rnd;
    for (;;) {
        QQ[q] = dg.lo;
        QQ[4 + q] = dg.hi;
        QQ[QS + q] = (q == 0) ? n_sent : 0u;
        QQ[QS + 4 + q] = 0u;
        n_sent++;
        rnd = b2_compress_quad<0>(QQ, x, hq, h4q, x.c0, x.d0 ^ dflag);
        uint32_t ok = (rnd.lo < cr.bound && rnd.hi < cr.bound) ? 1u : 0u;
        ok &= quad_perm<QROT1>(ok);
        ok &= quad_perm<QROT2>(ok);
        if (ok) break;
    }
    const uint32_t al = m31_reduce_2p(rnd.lo);
    const uint32_t k = cr.n_roots;
    tr->ch.digest[q] = dg.lo;
    tr->ch.digest[4 + q] = dg.hi;
    tr->alpha[q] = al;
    if (k < DT_MAX_LAYERS) {
        tr->roots[k][q] = root_lo;
        tr->roots[k][4 + q] = root_hi;
        tr->alphas[k][q] = al;
    }
    if (q == 0) {
        tr->ch.n_challenges = cr.n_chal + 1;
        tr->ch.n_sent = n_sent;
        tr->n_roots = k + 1;
    }
    cr.dg_lo = dg.lo;
    cr.dg_hi = dg.hi;
    cr.n_roots = k + 1;
    cr.n_chal += 1;
    return al;
}

// ------------------------------------------------------------------------------------------------
// tree9: nine levels per launch for level-A sizes 2^8 .. 2^17 (the launches that cannot fill the chip)
// ------------------------------------------------------------------------------------------------
// One workgroup owns 256 consecutive level-A nodes and takes them all the way to one hash: levels of 256 and 128 nodes one
// compression per lane (SoA LDS), the seven levels of 64 .. 1 nodes one compression per quad.  Such a launch is a pure
// latency chain, so what counts is the number of dependent compressions at ~1.8 us (lane) or ~1 us (quad) and the number
// of launches: a tree of 2^m leaves, m <= 17, is this launch plus the top kernel.
constexpr uint32_t T9_LEVELS = 9;

// Levels B .. (T9_LEVELS - 1) of a 256-node workgroup whose level-A hashes sit in RA (struct of arrays): 128 nodes one per lane,
// then 64 .. 1 nodes one per quad.  Shared by tree9_kernel and the fused small-domain kernel below.
__device__ __forceinline__ void tree9_upper(const TreeArgs& a, const uint32_t* RA, uint32_t* QQ, const QuadCtx& x, size_t wg_base, uint32_t t,
                                            uint32_t q, uint32_t quad) {
    // ---- level B (128 nodes): one per lane, results in the quad layout ----
    if (t < 128) {
        uint32_t m[16], h[8];
        lds_children(RA, 256 + 4, t, m);
        b2_merkle_block<B2_LAT>(m, h);
        if (a.store_all) store_hash(a.layers + layer_off(a.tree_log, a.level_a - 1), (wg_base >> 1) + t, h);
        q_put_hash(QQ, t, h);
    }
    __syncthreads();
    // ---- levels of 64, 32, ..., 1 nodes: one per quad ----
#pragma unroll
    for (uint32_t l = 2; l < T9_LEVELS; l += 2) {
        {
            const bool last = l + 1 == T9_LEVELS;
            uint8_t* gout = a.store_all ? a.layers + layer_off(a.tree_log, a.level_a - l) : (last ? a.last_out : nullptr);
            quad_level<0>(QQ, x, quad < (256u >> l), gout, (wg_base >> l) + quad, q, !last);
        }
        if (l + 1 < T9_LEVELS) {
            uint8_t* gout = a.store_all ? a.layers + layer_off(a.tree_log, a.level_a - (l + 1)) : nullptr;
            quad_level<1>(QQ, x, quad < (256u >> (l + 1)), gout, (wg_base >> (l + 1)) + quad, q, true);
        }
    }
}

template <int MODE, bool TP>
__global__ __launch_bounds__(256) void tree9_kernel(TreeArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t RA[8 * (256 + 4)];
    __shared__ __attribute__((aligned(16))) uint32_t QQ[2 * QBUF_WORDS];
    latency_kernel_priority();
    tree_args_of_blob(a);
    const uint32_t t = threadIdx.x, q = t & 3, quad = t >> 2;
    QuadCtx x;
    quad_ctx_init(x, quad, q);
    const size_t wg_base = (size_t)blockIdx.x * 256;  // the launcher guarantees 2^level_a >= 256
    // ---- level A (256 nodes): one per lane ----
    {
        const size_t g = wg_base + t;
        uint32_t h[8];
        if (MODE == T_NODE) {
            uint32_t m[16];
            load_children(a.children, g, m);
            b2_merkle_block<TP ? FRIEDA_B2_IDLE_NODE : B2_LAT>(m, h);
        } else if (MODE == T_LEAF4) {
            leaf_hash<TP ? FRIEDA_B2_IDLE_LEAF : B2_LAT>(a.cols[g], a.cols[a.col_stride + g], a.cols[2 * a.col_stride + g], a.cols[3 * a.col_stride + g], h);
        } else {
            const QM31Mat alpha = qm_matrix({a.tr->alpha[0], a.tr->alpha[1], a.tr->alpha[2], a.tr->alpha[3]});
            uint32_t it = (MODE == T_FOLD_CIRCLE) ? inv_circle_twiddle(a.itw, a.n, g, a.inv_init_y) : a.itw[g];
            QM31 r = fold_pair(a.cols, a.col_stride, g, it, alpha);
            a.out_vals[g] = r.a;
            a.out_vals[a.out_stride + g] = r.b;
            a.out_vals[2 * a.out_stride + g] = r.c;
            a.out_vals[3 * a.out_stride + g] = r.d;
            leaf_hash<TP ? FRIEDA_B2_IDLE_LEAF : B2_LAT>(r.a, r.b, r.c, r.d, h);
        }
        if (a.store_all && !a.skip_a) store_hash(a.layers + layer_off(a.tree_log, a.level_a), g, h);
        lds_put(RA, 256 + 4, t, h);
    }
    __syncthreads();
    tree9_upper(a, RA, QQ, x, wg_base, t, q, quad);
}

// ------------------------------------------------------------------------------------------------
// small domains: unpack + encode + leaf hashes + nine tree levels in ONE launch
// ------------------------------------------------------------------------------------------------
// The reference's own bench inputs are 1 KiB .. 64 KiB (benches/commit.rs:6-10, benches/proof.rs:14-21): a 1 KiB blob is 2^7
// coefficients per column on a 2^11 domain, and `commit` on it was four dependent launches of kernels shaped for megabytes (unpack 8.7
// us, transform 15.4, tree 14.9, top 7.6).  For domains of 2^8 .. 2^15 points and polynomials of <= 2^11 coefficients (blobs of <= 30 KB)
// a workgroup here owns 256 consecutive evaluations of the bit-reversed domain and does everything for them: it stages the blob in
// LDS, unpacks the 30-bit felts (src/utils.rs:10-33), runs the circle FFT of its block — a block of 2^L consecutive outputs is an
// independent size-2^L transform of the coefficient vector with that block's twiddles (ntt.hip, "Structure"); for L > 8 the L - 8
// layers that cross workgroups are computed in their one-output form (of each butterfly only the branch this workgroup's position
// selects: 2^(L-8) coefficients and one workgroup-uniform twiddle per layer and element) —, hashes its 256 leaves and reduces them to
// one hash exactly as tree9_kernel does.  The top kernel finishes the <= 128 hashes.  Same results, word for word, as the general path.
constexpr uint32_t SMALL_MAX_LOG_COEF = 11, SMALL_MIN_LOG_DOMAIN = 8, SMALL_MAX_LOG_DOMAIN = 15;

struct SmallFirstArgs {
    const uint8_t* data;  // blob bytes: device memory or page-locked host memory (read once per workgroup)
    size_t len, data_stride;
    uint32_t L, n, init_y;
    const uint32_t* tw;
    uint32_t* eval;  // non-null (generate_proof): the evaluation, 4 columns of 2^n words
    size_t eval_stride;
    TreeArgs tree;  // store_all / layers / tree_log / level_a / last_out / skip_a / bstride
};

namespace {
__device__ __forceinline__ uint32_t circle_twiddle_fwd(const uint32_t* __restrict__ tw, uint32_t h) {  // n >= 3: pairs (x, y) -> [y, -y, -x, x]
    const uint32_t j = h >> 2, r = h & 3u;
    const uint32_t v = tw[2 * j + (r < 2 ? 1 : 0)];
    return (r == 1 || r == 2) ? m31_neg(v) : v;
}

__global__ __launch_bounds__(256) void small_first_kernel(SmallFirstArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t SL[];  // small_first_lds_words(L): the blob, then the coefficients
    uint32_t* const BL = SL;                                        // the blob, zero beyond len, + a spare word: (15 << L) / 4 + 4 words
    uint32_t* const CO = SL + (((15u << a.L) + 3) / 4 + 4);         // coefficients, column c at c << L
    __shared__ __attribute__((aligned(16))) uint32_t V[4][256];
    __shared__ __attribute__((aligned(16))) uint32_t RA[8 * (256 + 4)];
    __shared__ __attribute__((aligned(16))) uint32_t QQ[2 * QBUF_WORDS];
    latency_kernel_priority();
    TreeArgs ta = a.tree;
    tree_args_of_blob(ta);
    const uint8_t* in = a.data + (size_t)blockIdx.y * a.data_stride;
    uint32_t* eval = a.eval ? reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(a.eval) + (size_t)blockIdx.y * a.tree.bstride) : nullptr;
    const uint32_t t = threadIdx.x, q = t & 3, quad = t >> 2, w = blockIdx.x;
    const uint32_t L = a.L, n = a.n;
    QuadCtx x;
    quad_ctx_init(x, quad, q);
    // ---- the bytes that can matter: 4 * 2^L felts of 30 bits ----
    const uint32_t need_w = ((15u << L) + 3) / 4;
    const size_t len = a.len;
    const bool al4 = (reinterpret_cast<uintptr_t>(in) & 3) == 0;
    for (uint32_t i = t; i < need_w; i += 256) {
        const size_t b = 4 * (size_t)i;
        uint32_t v = 0;
        if (al4 && b + 4 <= len) {
            v = reinterpret_cast<const uint32_t*>(in)[i];
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (b + k < len) v |= (uint32_t)in[b + k] << (8 * k);
        }
        BL[i] = v;
    }
    if (t == 0) BL[need_w] = 0;
    // ---- the twiddles of this thread's butterflies, requested before the first barrier: layer i pairs tile elements i0, i0 + 2^i ----
    const uint32_t top = L < 8 ? L : 8;  // layers 7 .. top of the block are replication (the coefficient vector is zero above 2^L)
    const uint32_t p = t & 127, cb = t >> 7;
    const uint32_t e = 256 * w + t;
    uint32_t twd[8], tup[3] = {0, 0, 0};  // tup[i - 8]: the (workgroup-uniform) twiddle of a layer i >= 8
#pragma unroll
    for (int i = 0; i < 8; i++) {
        twd[i] = 0;
        if ((uint32_t)i < top) {
            const uint32_t i0 = ((p >> i) << (i + 1)) | (p & ((1u << i) - 1));
            const uint32_t h = (256 * w + i0) >> (i + 1);
            twd[i] = i >= 1 ? a.tw[tw_level_offset_dev(n, (uint32_t)i - 1) + h] : circle_twiddle_fwd(a.tw, h);
        }
    }
#pragma unroll
    for (int i = 8; i < 11; i++)
        if ((uint32_t)i < L) tup[i - 8] = a.tw[tw_level_offset_dev(n, (uint32_t)i - 1) + (e >> (i + 1))];
    __syncthreads();
    // ---- bytes_to_felt_le: felt k = bits [30 k, 30 k + 30) ----
    for (uint32_t k = t; k < (4u << L); k += 256) {
        const uint32_t bit0 = 30 * k, wi = bit0 >> 5, sh = bit0 & 31;
        const uint64_t acc = (uint64_t)BL[wi] | ((uint64_t)BL[wi + 1] << 32);
        CO[k] = (uint32_t)(acc >> sh) & 0x3FFFFFFFu;
    }
    __syncthreads();
    // ---- element e of the block entering layer min(L, 8) - 1 ----
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint32_t* cc = CO + ((uint32_t)c << L);
        uint32_t v;
        if (L <= 8) {
            v = cc[t & ((1u << L) - 1)];
        } else {
            // layers L - 1 .. 8 pair e with positions other workgroups own.  Entering them, position p holds coefficient p mod 2^L; the
            // values our element depends on are those at e with its bits 8 .. L - 1 replaced by u: cc[t + 256 u].  Layer i keeps, of the
            // pair (u, u + 2^(i-8)), the branch bit i of e selects; its twiddle index e >> (i + 1) has only workgroup bits.
            uint32_t arr[8];
#pragma unroll
            for (int u = 0; u < 8; u++) arr[u] = (uint32_t)u < (1u << (L - 8)) ? cc[t + 256u * (uint32_t)u] : 0u;
#pragma unroll
            for (int i = 10; i >= 8; i--) {
                if ((uint32_t)i < L) {
                    const int half = 1 << (i - 8);
                    const bool hi = (w >> (i - 8)) & 1u;
#pragma unroll
                    for (int u = 0; u < half; u++) {
                        const uint32_t tt = m31_mul(arr[u + half], tup[i - 8]);
                        arr[u] = hi ? m31_sub(arr[u], tt) : m31_add(arr[u], tt);
                    }
                }
            }
            v = arr[0];
        }
        V[c][t] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if ((uint32_t)i < top) {  // uniform
            const uint32_t i0 = ((p >> i) << (i + 1)) | (p & ((1u << i) - 1)), i1 = i0 + (1u << i);
#pragma unroll
            for (int cc = 0; cc < 2; cc++) {
                uint32_t* col = V[cb + 2 * cc];
                const uint32_t x0 = col[i0], tt = m31_mul(col[i1], twd[i]);
                col[i0] = m31_add(x0, tt);
                col[i1] = m31_sub(x0, tt);
            }
            __syncthreads();
        }
    }
    // ---- leaves, then the nine levels of tree9 ----
    const uint32_t v0 = V[0][t], v1 = V[1][t], v2 = V[2][t], v3 = V[3][t];
    if (eval) {
        eval[e] = v0;
        eval[a.eval_stride + e] = v1;
        eval[2 * a.eval_stride + e] = v2;
        eval[3 * a.eval_stride + e] = v3;
    }
    {
        uint32_t h[8];
        leaf_hash<B2_LAT>(v0, v1, v2, v3, h);
        if (ta.store_all && !ta.skip_a) store_hash(ta.layers + layer_off(ta.tree_log, ta.level_a), e, h);
        lds_put(RA, 256 + 4, t, h);
    }
    __syncthreads();
    tree9_upper(ta, RA, QQ, x, (size_t)256 * w, t, q, quad);
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// tree7q: the narrow middle of a tree (<= 32768 level-A nodes) hashed by quads across many workgroups
// ------------------------------------------------------------------------------------------------
// One workgroup (64 quads) owns 64 consecutive level-A nodes and produces up to seven levels (64, 32, ..., 1 nodes), every
// node by the 4-lane compression: ~1 us per level instead of ~1.8 us for one-hash-per-lane, and the wide levels spread
// over many CUs instead of queueing in the single-workgroup top kernel.
constexpr uint32_t T7Q_UNITS = 64;
constexpr uint32_t T7Q_LEVELS = 7;

__global__ __launch_bounds__(256) void tree7q_kernel(TreeArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t QQ[2 * QBUF_WORDS];
    latency_kernel_priority();
    tree_args_of_blob(a);
    const uint32_t t = threadIdx.x, q = t & 3, quad = t >> 2;
    QuadCtx x;
    quad_ctx_init(x, quad, q);
    const size_t total_a = (size_t)1 << a.level_a;
    const size_t wg_base = (size_t)blockIdx.x * T7Q_UNITS;
    const uint32_t cnt_a = (uint32_t)(total_a - wg_base < T7Q_UNITS ? total_a - wg_base : T7Q_UNITS);
    uint32_t nl = 1;
    while (nl < T7Q_LEVELS && (cnt_a >> nl) >= 1) nl++;
    // the 2 * cnt_a children of this workgroup's level-A nodes: global (array of structs) -> quad layout, buffer 0
    q_load_hashes(QQ, a.children + 64 * wg_base, 2 * cnt_a, t, 256);
    __syncthreads();
    uint32_t l = 0;
    while (l < nl) {
        {
            const bool last = l + 1 == nl;
            uint8_t* gout = a.store_all ? a.layers + layer_off(a.tree_log, a.level_a - l) : (last ? a.last_out : nullptr);
            quad_level<0>(QQ, x, quad < (cnt_a >> l), gout, (wg_base >> l) + quad, q, !last);
        }
        if (++l >= nl) break;
        {
            const bool last = l + 1 == nl;
            uint8_t* gout = a.store_all ? a.layers + layer_off(a.tree_log, a.level_a - l) : (last ? a.last_out : nullptr);
            quad_level<1>(QQ, x, quad < (cnt_a >> l), gout, (wg_base >> l) + quad, q, !last);
        }
        ++l;
    }
}

struct TopArgs {
    const uint8_t* in;  // 2^l_in hashes (array of structs)
    uint32_t l_in;      // <= 11 supported; build_tree hands over at <= 9
    uint8_t* layers;    // non-null: store every produced level at its leaves-first offset
    uint32_t tree_log;
    uint8_t* root_out;  // non-null: also store the root here
    DevTranscript* tr;  // non-null: mix the root and draw the next alpha
    size_t bstride;     // batch: bytes between blobs' workspaces (blob = blockIdx.y); tr is an array
    // non-null (first tree of a proof): the initial transcript, read straight from pinned host memory at kernel start (blob b at
    // tr_init + b * tr_init_pitch bytes) — the device transcript is initialised from it here instead of by a copy in the stream
    const DevTranscript* tr_init;
    size_t tr_init_pitch;
};

__global__ __launch_bounds__(WG1_THREADS) void top_kernel(TopArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t S0[8 * (1024 + 4)];
    __shared__ __attribute__((aligned(16))) uint32_t S1[8 * (512 + 4)];
    __shared__ __attribute__((aligned(16))) uint32_t QQ[2 * QBUF_WORDS];
    latency_kernel_priority();
    {
        const size_t off = (size_t)blockIdx.y * a.bstride;
        a.in += off;
        if (a.layers) a.layers += off;
        if (a.root_out) a.root_out += off;
        if (a.tr) a.tr += blockIdx.y;
    }
    const uint32_t t = threadIdx.x, q = t & 3;
    QuadCtx x;
    quad_ctx_init(x, t >> 2, q);
    ChanRegs cr = {};
    if (a.tr && t < 4) {
        if (a.tr_init) {
            const DevTranscript* src =
                reinterpret_cast<const DevTranscript*>(reinterpret_cast<const char*>(a.tr_init) + (size_t)blockIdx.y * a.tr_init_pitch);
            chan_prefetch(cr, src, q);
            if (q == 0) a.tr->status = 0;
            if (q == 1) a.tr->nonce = ~0ull;
            if (q == 2) a.tr->n_last_poly = 0;
            if (q == 3) a.tr->draw_bound = cr.bound;
        } else {
            chan_prefetch(cr, a.tr, q);
        }
    }
    const uint32_t* root_lds;
    if (a.l_in == 0) {
        if (t < 8) QQ[t] = reinterpret_cast<const uint32_t*>(a.in)[t];
        __syncthreads();
        root_lds = QQ;
    } else {
        const uint32_t l = a.l_in - 1, cnt = 1u << l;
        uint8_t* gout = a.layers ? a.layers + layer_off(a.tree_log, l) : nullptr;
        if (cnt >= 256) {
            for (uint32_t j = t; j < cnt; j += WG1_THREADS) {
                uint32_t m[16], h[8];
                load_children(a.in, j, m);
                b2_merkle_block<B2_LAT>(m, h);
                if (gout) store_hash(gout, j, h);
                if (cnt >= 512)
                    lds_put(S0, cnt + 4, j, h);
                else
                    q_put_hash(QQ, j, h);
            }
            __syncthreads();
            root_lds = wg_reduce(S0, S1, QQ, l, a.layers, a.tree_log, x);
        } else {
            // children (2 cnt <= 256 hashes) from global into the quad layout, then quad levels all the way
            q_load_hashes(QQ, a.in, 2 * cnt, t, WG1_THREADS);
            __syncthreads();
            root_lds = wg_reduce(S0, S1, QQ, l + 1, a.layers, a.tree_log, x);
        }
    }
    if (t < 4) {
        const uint32_t root_lo = root_lds[q], root_hi = root_lds[4 + q];
        if (a.root_out) {
            uint32_t* ro = reinterpret_cast<uint32_t*>(a.root_out);
            ro[q] = root_lo;
            ro[4 + q] = root_hi;
        }
        if (a.tr) channel_after_root_quad(a.tr, cr, root_lo, root_hi, QQ, x);
    }
}

// ------------------------------------------------------------------------------------------------
// FRI tail: every remaining layer of <= 2048 points in one workgroup
// ------------------------------------------------------------------------------------------------
constexpr uint32_t TAIL_MAX_LAYERS = 16;
constexpr uint32_t TAIL_CAP = 2048;

struct TailArgs {
    const uint32_t* src;  // current layer (its tree is committed and the alpha for folding it is in tr)
    size_t src_stride;
    uint32_t src_log;
    int src_is_circle;
    uint32_t n;
    const uint32_t* itw;
    uint32_t inv_init_y;
    uint32_t last_log;  // log size of the last layer's domain
    uint32_t last;      // log_last_layer_degree_bound
    uint32_t n_layers;  // folds to perform; the last one produces the last layer
    uint32_t* vals[TAIL_MAX_LAYERS];
    uint8_t* trees[TAIL_MAX_LAYERS];
    DevTranscript* tr;
    size_t bstride;   // batch: bytes between blobs' workspaces (blob = blockIdx.y); tr is an array
    uint32_t* gnext;  // non-null: the grind's per-blob window counters, zeroed here for the grind launch that follows
};

__global__ __launch_bounds__(WG1_THREADS) void tail_kernel(TailArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t S0[8 * (TAIL_CAP + 4)];
    __shared__ __attribute__((aligned(16))) uint32_t S1[8 * (TAIL_CAP / 2 + 4)];
    __shared__ __attribute__((aligned(16))) uint32_t QQ[2 * QBUF_WORDS];
    __shared__ uint32_t s_alpha[4];
    latency_kernel_priority();
    uint32_t* const MSG = S0 + 4 * TAIL_CAP + 8;  // the upper half of S0 is free once the last layer is reached
    const uint32_t t = threadIdx.x, q = t & 3;
    const size_t boff = (size_t)blockIdx.y * a.bstride;  // this workgroup's blob
    DevTranscript* tr = a.tr + blockIdx.y;
    QuadCtx x;
    quad_ctx_init(x, t >> 2, q);
    ChanRegs cr = {};
    if (t < 4) {
        chan_prefetch(cr, tr, q);
        s_alpha[t] = tr->alpha[t];
    }
    if (t == 4 && a.gnext) a.gnext[blockIdx.y * GRIND_NEXT_STRIDE] = 0;

    const uint32_t* src = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.src) + boff);
    size_t src_stride = a.src_stride;
    uint32_t src_log = a.src_log;
    bool circle = a.src_is_circle != 0;

    for (uint32_t kx = 0; kx < a.n_layers; kx++) {
        const uint32_t m_new = src_log - 1, cnt = 1u << m_new;
        const bool is_last = kx + 1 == a.n_layers;
        __syncthreads();  // s_alpha of this layer (and, from the second layer on, the previous layer's values) are in place
        const QM31Mat alpha = qm_matrix({s_alpha[0], s_alpha[1], s_alpha[2], s_alpha[3]});
        // line layer of log size src_log sits on twiddle level n - 1 - src_log
        const uint32_t* itw_level = circle ? a.itw : a.itw + tw_level_offset_dev(a.n, a.n - 1 - src_log);
        uint32_t* dstv = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(a.vals[kx]) + boff);
        for (uint32_t j = t; j < cnt; j += WG1_THREADS) {
            uint32_t it = circle ? inv_circle_twiddle(a.itw, a.n, j, a.inv_init_y) : itw_level[j];
            QM31 r = fold_pair(src, src_stride, j, it, alpha);
            dstv[j] = r.a;
            dstv[cnt + j] = r.b;
            dstv[2 * cnt + j] = r.c;
            dstv[3 * cnt + j] = r.d;
            if (!is_last) {
                uint32_t h[8];
                leaf_hash<B2_LAT>(r.a, r.b, r.c, r.d, h);
                store_hash(a.trees[kx] + boff, j, h);  // leaf layer sits at offset 0
                if (cnt >= 512)
                    lds_put(S0, cnt + 4, j, h);
                else
                    q_put_hash(QQ, j, h);
            } else {
                // keep the last layer in LDS for the interpolation: coordinate c of point j at S0[c * cnt + j]
                S0[j] = r.a;
                S0[cnt + j] = r.b;
                S0[2 * cnt + j] = r.c;
                S0[3 * cnt + j] = r.d;
            }
        }
        __syncthreads();
        if (is_last) break;
        const uint32_t* root_lds = wg_reduce(S0, S1, QQ, m_new, a.trees[kx] + boff, m_new, x);
        if (t < 4) {
            const uint32_t root_lo = root_lds[q], root_hi = root_lds[4 + q];
            s_alpha[t] = channel_after_root_quad(tr, cr, root_lo, root_hi, QQ, x);
        }
        src = dstv;
        src_stride = cnt;
        src_log = m_new;
        circle = false;
    }

    // ---- FriProver::commit_last_layer: LineEvaluation::interpolate (stwo core/poly/line.rs) ----
    const uint32_t lg = a.last_log, cnt = 1u << lg;
    // bit-reverse into natural order: W[c][i] = V[c][brev(i)]
    uint32_t* V = S0;
    uint32_t* W = S1;
    for (uint32_t i = t; i < cnt; i += WG1_THREADS) {
        uint32_t b = bit_reverse(i, lg);
#pragma unroll
        for (int c = 0; c < 4; c++) W[c * cnt + i] = V[c * cnt + b];
    }
    __syncthreads();
    // line_ifft: for sub-domain size 2^k (k = lg .. 1): (l, r) -> (l + r, (l - r) / x_i), x_i = the i-th point of the
    // k-log coset in natural order = inverse level n-1-k at index brev(i, k-1)
    for (uint32_t kk = lg; kk >= 1; kk--) {
        const uint32_t half = 1u << (kk - 1);
        const uint32_t* lvl = a.itw + tw_level_offset_dev(a.n, a.n - 1 - kk);
        for (uint32_t p = t; p < (cnt >> 1); p += WG1_THREADS) {
            uint32_t i = p & (half - 1), base = (p >> (kk - 1)) << kk;
            uint32_t it = lvl[bit_reverse(i, kk - 1)];
            uint32_t i0 = base + i, i1 = i0 + half;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                uint32_t l = W[c * cnt + i0], r = W[c * cnt + i1];
                W[c * cnt + i0] = m31_add(l, r);
                W[c * cnt + i1] = m31_mul(m31_sub(l, r), it);
            }
        }
        __syncthreads();
    }
    // scale by 1/len = 2^(31 - lg) (2^31 == 1 mod P); W is now the coefficient vector in LinePoly's internal order.
    // Ordered coefficient brev(j) must vanish for brev(j) >= 2^last; the kept coefficients, re-bit-reversed over `last`
    // bits, are W[q << (lg - last)].
    const uint32_t len_inv = (lg == 0) ? 1u : (1u << (31 - lg));
    const uint32_t sh = lg - a.last;
    uint32_t bad = 0;
    for (uint32_t j = t; j < cnt; j += WG1_THREADS) {
        uint32_t v[4];
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = m31_mul(W[c * cnt + j], len_inv);
        if ((j & ((1u << sh) - 1)) == 0) {
            uint32_t qq = j >> sh;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                MSG[8 + 4 * qq + c] = v[c];
                tr->last_poly[4 * qq + c] = v[c];
            }
        } else if (v[0] | v[1] | v[2] | v[3]) {
            bad = 1;
        }
    }
    if (bad) atomicOr(&tr->status, 1u);
    const uint32_t n_poly = 1u << a.last;
    const uint32_t msg_words = 8 + 4 * n_poly;
    for (uint32_t i = msg_words + t; i < ((msg_words + 15) / 16) * 16; i += WG1_THREADS) MSG[i] = 0;
    __syncthreads();
    if (t == 0) {
        // Blake2sChannel::mix_felts(last_layer_poly)
        Channel ch = tr->ch;
#pragma unroll
        for (int w = 0; w < 8; w++) MSG[w] = ch.digest[w];
        uint32_t r[8];
        b2s256_words(MSG, 4 * msg_words, r);
        ch.update_digest(r);
        tr->ch = ch;
        tr->n_last_poly = n_poly;
    }
}

// ------------------------------------------------------------------------------------------------
// grind: proof of work over the digests in the device transcripts, one work queue per blob
// ------------------------------------------------------------------------------------------------
// The nonces base .. base + n_windows * GRIND_WINDOW of every blob are cut into windows that workgroups claim in increasing
// order from the blob's counter (`next`, zeroed before the launch).  A workgroup walks round-robin over the blobs that are
// still searching (wave 0 tests 64 blobs per step), claims the next window of one, scans it (one compression per nonce,
// atomicMin on a hit) and moves on; it leaves once no blob has work.  Windows are claimed in order and always scanned to the
// point where a smaller hit is known, so tr->nonce ends as the MINIMUM qualifying nonce of the range, as in the reference's
// sequential search; blobs that finish early release their workgroups to the others (in a batch the slowest blob needs
// several times the mean).  Every loop is bounded: a claim consumes one of n_windows * batch windows, a fruitless walk over
// all blobs ends the workgroup.
constexpr uint32_t GRIND_WINDOW = 1024;  // the unit n_windows counts in: 4 nonces per lane (256-nonce windows were tried: the claim — an atomic
                                         // and two barriers — then costs as much as the scan, 47 instead of 19 us on the bench blob)
// A claim takes `iters / 4` consecutive units (GrindArgs::iters nonces per lane): in a batch 2048 workgroups claim ~45 M units per second
// chip-wide, and with every blob's counter in the same cache line and 4 nonces per lane per claim the launch ran at 36 - 50 % of the
// compression rate (10 blobs of 2^24: 772 us for 12.9 M nonces; 32 blobs of 2^20: 1089 us for 25.2 M).  Round 5: a cache line per
// counter (GRIND_NEXT_STRIDE: 637 / 397 us) and 8 nonces per lane and claim for batches of 2 .. 15 blobs (337 us).

struct GrindArgs {
    DevTranscript* tr;  // array over the blobs of the batch
    uint32_t* next;     // [batch] next unclaimed window of each blob
    uint32_t pow_bits, batch;
    unsigned long long base;
    uint32_t n_windows;  // per blob, in units of GRIND_WINDOW nonces
    uint32_t units;      // units per claim (a lane scans 4 * units nonces of the claim)
};

__global__ __launch_bounds__(256) void grind_dev_kernel(GrindArgs a) {
    __shared__ uint32_t s_claim[2];  // blob, window (blob == ~0: nothing left)
    const uint32_t t = threadIdx.x;
    uint32_t b = blockIdx.x % a.batch;  // wave 0's walking position
    for (;;) {
        if (t < 64) {
            uint32_t got_b = ~0u, got_w = 0, idle = 0;
            while (idle < a.batch) {
                uint32_t bb = b + t;
                if (bb >= a.batch) bb -= a.batch;
                const bool cand = t < a.batch && __hip_atomic_load(&a.tr[bb].nonce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ~0ull &&
                                  __hip_atomic_load(&a.next[bb * GRIND_NEXT_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.n_windows;
                const unsigned long long mask = __ballot(cand);
                if (mask == 0) {
                    const uint32_t adv = a.batch < 64 ? a.batch : 64;
                    b += adv;
                    if (b >= a.batch) b -= a.batch;
                    idle += adv;
                    continue;
                }
                const uint32_t l = (uint32_t)__ffsll((long long)mask) - 1;
                uint32_t tb = b + l;
                if (tb >= a.batch) tb -= a.batch;
                uint32_t c = 0;
                if (t == 0) c = atomicAdd(&a.next[tb * GRIND_NEXT_STRIDE], a.units);
                c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                b = tb + 1 == a.batch ? 0 : tb + 1;
                if (c < a.n_windows) {
                    got_b = tb;
                    got_w = c;
                    break;
                }
                idle += l + 1;  // lost the race for that blob's last window: keep walking
            }
            if (t == 0) {
                s_claim[0] = got_b;
                s_claim[1] = got_w;
            }
        }
        __syncthreads();
        const uint32_t cb = s_claim[0], cw = s_claim[1];
        __syncthreads();
        if (cb == ~0u) return;
        DevTranscript* tr = a.tr + cb;
        uint32_t h[8], pre[14];
#pragma unroll
        for (int i = 0; i < 8; i++) h[i] = tr->ch.digest[i];
        b2_grind_prepare(h, pre);  // the part of round 0 that does not see the nonce: once per claim (blake2s.h)
        // trailing_zeros >= pow_bits needs the low min(pow_bits, 32) bits of word 0 clear: decided from word 0 alone; a survivor (one nonce
        // in 2^pow_bits) is recomputed in full below
        const uint32_t low_mask = a.pow_bits >= 32 ? 0xFFFFFFFFu : ((1u << a.pow_bits) - 1u);
        const unsigned long long first = a.base + (unsigned long long)cw * GRIND_WINDOW + t;
        // (the claim may reach past the range's last unit: only whole units below n_windows are scanned)
        const uint32_t my_units = a.n_windows - cw < a.units ? a.n_windows - cw : a.units;
        for (uint32_t i = 0; i < my_units * (GRIND_WINDOW / 256); i++) {
            const unsigned long long nonce = first + 256ull * i;
            // a smaller qualifying nonce is already known: nothing this lane finds from here on can lower the minimum
            if (__hip_atomic_load(&tr->nonce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nonce) break;
            if ((b2_grind_word0(h, pre, (uint32_t)nonce, (uint32_t)(nonce >> 32)) & low_mask) != 0) continue;  // (a chip-filling launch: the throughput form)
            const uint32_t m[16] = {(uint32_t)nonce, (uint32_t)(nonce >> 32), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            uint32_t r[8];
            b2_compress(h, m, 0, 0, 0, 0, r);  // the survivor, in full (plain form: one lane in 2^pow_bits gets here)
            uint32_t tz;
            if (r[0])
                tz = __ffs(r[0]) - 1;
            else if (r[1])
                tz = 32 + __ffs(r[1]) - 1;
            else if (r[2])
                tz = 64 + __ffs(r[2]) - 1;
            else if (r[3])
                tz = 96 + __ffs(r[3]) - 1;
            else
                tz = 128;
            if (tz >= a.pow_bits) atomicMin(&tr->nonce, nonce);
        }
    }
}

// Which kernel produces level A (2^level_a nodes per blob, `batch` blobs per launch) and how many levels it yields:
//   T_WIDE   tree5r, 1024-node workgroups, five levels: >= 2^18 nodes in the launch (all blobs together), >= 1024 per blob
//            (alternating A/B runs: 2^18 and 2^19 launches are ~10 us faster here than with 256-node workgroups)
//   T_NINE   tree9, 256-node workgroups, nine levels: 2^8 .. 2^17 nodes per blob — a latency chain, fewest launches
//   T_SMALL  tree5<256>, 256-node workgroups, five levels: everything else (unaligned Level B buffers; < 2^8 nodes: a partial
//            workgroup)
enum TreeKernel { T_WIDE, T_WIDE3, T_NINE, T_SMALL };
// the compression's throughput form pays from ~3 waves per SIMD on (256-thread workgroups: 3 per CU); latency-shaped launches keep
// the plain form (microbenchmarks: no gain at 2 waves per SIMD, +12 % latency for a lone wave; tests/cpp/bench_api: 16 KiB .. 256 KiB
// lone proofs 2 - 3 % slower, a lone 2^20 proof 8 % slower with the throughput form in every launch)
bool tp_launch(const Tuning& tn, size_t workgroups) { return workgroups >= tn.tp_min_wgs; }
constexpr uint32_t T9_MIN_LOG = 8;
// tuning knobs (kernels.h Tuning: defaults = measured best): smallest launch of the register-subtree kernel; three-register-level form;
// largest level-A size of the nine-level kernel; largest hand-over size of the top kernel
TreeKernel tree_kernel_for(const Tuning& tn, uint32_t level_a, uint32_t batch, bool aligned16, bool reg3_ok = false) {
    uint32_t batch_log = 0;
    while ((2u << batch_log) <= batch) batch_log++;
    if (reg3_ok && aligned16 && tn.t5_reg3_log && level_a >= 10 && level_a + batch_log >= tn.t5_reg3_log) return T_WIDE3;
    // the register-subtree kernel uses 16-byte column accesses; anything unaligned (Level B callers may pass any pointers)
    // takes a 256-unit kernel, which produces the same hashes
    if (aligned16 && level_a >= 10 && level_a + batch_log >= tn.t5_wide_log) return T_WIDE;
    if (level_a >= T9_MIN_LOG && level_a <= tn.t9_max_log) return T_NINE;
    return T_SMALL;
}
uint32_t tree_kernel_levels(TreeKernel k, uint32_t level_a) {
    if (k == T_NINE) return T9_LEVELS;
    if (k == T_WIDE3) return 3;
    const uint32_t in_wg = level_a < (k == T_WIDE ? 10u : 8u) ? level_a : (k == T_WIDE ? 10u : 8u);  // log2 of a workgroup's A nodes
    return in_wg + 1 < T5_LEVELS ? in_wg + 1 : T5_LEVELS;
}
bool tree_args_aligned16(const TreeArgs& a) {
    return ((reinterpret_cast<uintptr_t>(a.cols) | reinterpret_cast<uintptr_t>(a.out_vals) | reinterpret_cast<uintptr_t>(a.itw) |
             (a.col_stride * 4) | (a.out_stride * 4)) & 15) == 0;
}

// launches level A (+ the levels its kernel yields) and returns the number of levels produced
uint32_t launch_tree_a(const Launch& L, int mode, const TreeArgs& a, const char* name, double (*bytes_of)(int, uint32_t, uint32_t)) {
    const size_t total = (size_t)1 << a.level_a;
    // the three-level form only where every level is kept (its last level is four times what the root-only scratch holds)
    const TreeKernel k = tree_kernel_for(*L.tune, a.level_a, L.batch, tree_args_aligned16(a), mode != T_NODE && a.store_all);
    const uint32_t levels = tree_kernel_levels(k, a.level_a);
    const uint32_t units = (k == T_WIDE || k == T_WIDE3) ? T5_UNITS : 256u;
    const dim3 grid((unsigned)((total + units - 1) / units), L.batch);
    Scope scope(L, name, bytes_of(mode, a.level_a, levels));
    const bool tp = tp_launch(*L.tune, (size_t)grid.x * grid.y);
    if (k == T_SMALL) {
        switch (mode) {
            case T_LEAF4: if (tp) tree5_kernel<T_LEAF4, 256, true><<<grid, T5_THREADS, 0, L.stream>>>(a); else tree5_kernel<T_LEAF4, 256, false><<<grid, T5_THREADS, 0, L.stream>>>(a); break;
            case T_NODE: if (tp) tree5_kernel<T_NODE, 256, true><<<grid, T5_THREADS, 0, L.stream>>>(a); else tree5_kernel<T_NODE, 256, false><<<grid, T5_THREADS, 0, L.stream>>>(a); break;
            case T_FOLD_CIRCLE: if (tp) tree5_kernel<T_FOLD_CIRCLE, 256, true><<<grid, T5_THREADS, 0, L.stream>>>(a); else tree5_kernel<T_FOLD_CIRCLE, 256, false><<<grid, T5_THREADS, 0, L.stream>>>(a); break;
            default: if (tp) tree5_kernel<T_FOLD_LINE, 256, true><<<grid, T5_THREADS, 0, L.stream>>>(a); else tree5_kernel<T_FOLD_LINE, 256, false><<<grid, T5_THREADS, 0, L.stream>>>(a); break;
        }
    } else if (k == T_NINE) {
        switch (mode) {
            case T_LEAF4: if (tp) tree9_kernel<T_LEAF4, true><<<grid, 256, 0, L.stream>>>(a); else tree9_kernel<T_LEAF4, false><<<grid, 256, 0, L.stream>>>(a); break;
            case T_NODE: if (tp) tree9_kernel<T_NODE, true><<<grid, 256, 0, L.stream>>>(a); else tree9_kernel<T_NODE, false><<<grid, 256, 0, L.stream>>>(a); break;
            case T_FOLD_CIRCLE: if (tp) tree9_kernel<T_FOLD_CIRCLE, true><<<grid, 256, 0, L.stream>>>(a); else tree9_kernel<T_FOLD_CIRCLE, false><<<grid, 256, 0, L.stream>>>(a); break;
            default: if (tp) tree9_kernel<T_FOLD_LINE, true><<<grid, 256, 0, L.stream>>>(a); else tree9_kernel<T_FOLD_LINE, false><<<grid, 256, 0, L.stream>>>(a); break;
        }
    } else if (k == T_WIDE3) {
        switch (mode) {
            case T_LEAF4: if (tp) tree5r_kernel<T_LEAF4, true, true><<<grid, T5_THREADS, 0, L.stream>>>(a); else tree5r_kernel<T_LEAF4, true, false><<<grid, T5_THREADS, 0, L.stream>>>(a); break;
            case T_FOLD_CIRCLE: if (tp) tree5r_kernel<T_FOLD_CIRCLE, true, true><<<grid, T5_THREADS, 0, L.stream>>>(a); else tree5r_kernel<T_FOLD_CIRCLE, true, false><<<grid, T5_THREADS, 0, L.stream>>>(a); break;
            default: if (tp) tree5r_kernel<T_FOLD_LINE, true, true><<<grid, T5_THREADS, 0, L.stream>>>(a); else tree5r_kernel<T_FOLD_LINE, true, false><<<grid, T5_THREADS, 0, L.stream>>>(a); break;
        }
    } else {
        switch (mode) {
            case T_LEAF4: if (tp) tree5r_kernel<T_LEAF4, false, true><<<grid, T5_THREADS, 0, L.stream>>>(a); else tree5r_kernel<T_LEAF4, false, false><<<grid, T5_THREADS, 0, L.stream>>>(a); break;
            case T_NODE: if (tp) tree5r_kernel<T_NODE, false, true><<<grid, T5_THREADS, 0, L.stream>>>(a); else tree5r_kernel<T_NODE, false, false><<<grid, T5_THREADS, 0, L.stream>>>(a); break;
            case T_FOLD_CIRCLE: if (tp) tree5r_kernel<T_FOLD_CIRCLE, false, true><<<grid, T5_THREADS, 0, L.stream>>>(a); else tree5r_kernel<T_FOLD_CIRCLE, false, false><<<grid, T5_THREADS, 0, L.stream>>>(a); break;
            default: if (tp) tree5r_kernel<T_FOLD_LINE, false, true><<<grid, T5_THREADS, 0, L.stream>>>(a); else tree5r_kernel<T_FOLD_LINE, false, false><<<grid, T5_THREADS, 0, L.stream>>>(a); break;
        }
    }
    return levels;
}

constexpr uint32_t T7Q_MAX_LEVEL_A = 15;  // level-A sizes up to 2^15 nodes go through tree7q

// algorithmic bytes of `levels` consecutive node levels whose first (largest) has 2^la nodes: 64 B in + 32 B out each
double node_levels_bytes(uint32_t la, uint32_t levels) {
    double b = 0;
    for (uint32_t i = 0; i < levels; i++) b += 96.0 * (double)((size_t)1 << (la - i));
    return b;
}

}  // namespace

FR_CLOCK_READER(frieda_debug_clock_tree5r, g_clock_tree5r)

// The part of a tree above an already produced level: level `cur` (2^cur hashes per blob) sits at cur_ptr (inside `a.layers` at
// its leaves-first offset when every level is kept, else in scratch half s0 or s1); node launches until the single-workgroup top
// kernel can take over, which finishes with the root (and the channel step when tr).
void finish_tree(const Launch& L, const TreeArgs& a, uint32_t m, uint32_t cur, const uint8_t* cur_ptr, uint8_t* s0, uint8_t* s1,
                 uint8_t* root_out, DevTranscript* tr, const DevTranscript* tr_init, size_t tr_init_pitch) {
    uint8_t* layers = a.store_all ? a.layers : nullptr;
    auto bytes_of = [](int, uint32_t la, uint32_t levels) -> double { return node_levels_bytes(la, levels); };
    while (cur > L.tune->top_max_log) {
        TreeArgs b = a;
        b.skip_a = 0;
        b.skip_bc = 0;
        b.level_a = cur - 1;
        b.children = cur_ptr;
        b.last_out = (cur_ptr == s0) ? s1 : s0;
        uint32_t l2;
        if (cur - 1 <= T7Q_MAX_LEVEL_A) {
            // narrow: quads over many workgroups, up to seven levels
            const uint32_t in_wg = b.level_a < 6 ? b.level_a : 6;
            l2 = in_wg + 1 < T7Q_LEVELS ? in_wg + 1 : T7Q_LEVELS;
            Scope scope(L, "tree7q_node", node_levels_bytes(cur - 1, l2));
            const dim3 grid((unsigned)((((size_t)1 << b.level_a) + T7Q_UNITS - 1) / T7Q_UNITS), L.batch);
            tree7q_kernel<<<grid, 256, 0, L.stream>>>(b);
        } else {
            l2 = launch_tree_a(L, T_NODE, b, "tree5_node", bytes_of);
        }
        cur = cur - l2;
        cur_ptr = layers ? layers + merkle_layer_offset(m, cur) : b.last_out;
    }
    TopArgs tp;
    tp.in = cur_ptr;
    tp.l_in = cur;
    tp.layers = layers;
    tp.tree_log = m;
    tp.root_out = root_out;
    tp.tr = tr;
    tp.bstride = L.bstride;
    tp.tr_init = tr_init;
    tp.tr_init_pitch = tr_init_pitch;
    {
        Scope scope(L, "tree_top", node_levels_bytes(cur > 0 ? cur - 1 : 0, cur));
        top_kernel<<<dim3(1, L.batch), WG1_THREADS, 0, L.stream>>>(tp);
    }
}

// Builds the tree of a layer whose level A is produced by `mode`; finishes with the root (and the channel step when tr).
// `layers` non-null = keep every level (leaves-first); else only the root survives and `scratch` (>= 2 * 32 * 2^(m-4) B) is used.
void build_tree(const Launch& L, int mode, TreeArgs a, uint32_t m, uint8_t* layers, uint8_t* scratch, uint8_t* root_out,
                DevTranscript* tr, const DevTranscript* tr_init = nullptr, size_t tr_init_pitch = 0) {
    a.level_a = m;
    a.tree_log = m;
    a.layers = layers;
    a.store_all = layers != nullptr;
    a.skip_bc = a.store_all && a.skip_a && m >= tree_skip_threshold(*L.tune, L.batch) && mode != T_NODE;  // (level A == the leaves of this tree)
    a.bstride = L.bstride;
    uint8_t* s0 = scratch;
    uint8_t* s1 = scratch ? scratch + ((size_t)32 << (m > 4 ? m - 4 : 0)) : nullptr;
    a.last_out = s0;
    // level A: leaves (16 B columns in, 32 B out) or fold+leaves (32 B pair in, 16 B values + 32 B hash out); nodes: 96 B each
    auto bytes_of = [](int md, uint32_t la, uint32_t levels) -> double {
        if (md == T_NODE) return node_levels_bytes(la, levels);
        return (md == T_LEAF4 ? 48.0 : 80.0) * (double)((size_t)1 << la) + node_levels_bytes(la > 0 ? la - 1 : 0, levels - 1);
    };
    const char* nm = mode == T_LEAF4 ? "tree5_leaf" : (mode == T_FOLD_CIRCLE ? "tree5_fold_circle" : "tree5_fold_line");
    const uint32_t lv = launch_tree_a(L, mode, a, nm, bytes_of);
    const uint32_t cur = m - (lv - 1);  // lowest-index (smallest) level produced so far
    finish_tree(L, a, m, cur, layers ? layers + merkle_layer_offset(m, cur) : s0, s0, s1, root_out, tr, tr_init, tr_init_pitch);
}

void encode_and_first_tree(const Launch& L, const uint32_t* d_coef, size_t coef_stride, uint32_t Lc, uint32_t n, const uint32_t* d_tw,
                           DomainScalars ds, uint32_t* d_eval, size_t eval_stride, uint8_t* d_layers, uint8_t* d_scratch, uint8_t* d_root,
                           DevTranscript* tr, const DevTranscript* tr_init, size_t tr_init_pitch) {
    // The fused last pass hashes a whole 4096-leaf subtree per workgroup: with fewer tiles than CUs (a lone blob of <= 256 KB) most of
    // the chip idles while 32 - 128 workgroups each run ~32 compressions per thread; the separate transform + 256-leaf tree launches
    // spread the same work over 16 times the workgroups (commit of 64 KiB: 123 -> 76 us, of 256 KiB: 133 -> 121; from 256 tiles on the
    // fused launch is ahead again: 120 vs 136 us at 2^20).  Batches count: 64 blobs of 64 KiB are 2048 tiles.
    // With every level KEPT (a proof) the two launches win at every size since the compressions switch wave priority (blake2s.h): the
    // tree launch runs 8 waves per SIMD, the fused one 4 (120 VGPRs), and a compression costs 2390 against 2830 cycles there (stream of
    // 2^24 proofs 1.50 vs 1.55 ms, 2^22 0.404 vs 0.415, lone 2^21 0.655 vs 0.680; profiles/r05_prio_product_ab.txt).  A commitment
    // (nothing kept) stays fused: the evaluations are never written (lone 2^24 0.785 vs 0.830 ms, stream 0.77 vs 0.85).
    const bool no_fuse = L.tune->no_encode_tree_fusion /* A/B knob */ || (((size_t)1 << (n > 12 ? n - 12 : 0)) * L.batch < 256) ||
                         (d_layers != nullptr && !L.tune->encode_tree_fusion_prove /* A/B knob */);
    uint8_t* s0 = d_scratch;
    uint8_t* s1 = d_scratch ? d_scratch + ((size_t)32 << (n > 4 ? n - 4 : 0)) : nullptr;
    EncodeTreeSink sink{d_layers, s0};
    const uint32_t fused = circle_evaluate_into_tree(L, d_coef, coef_stride, 4, Lc, n, d_tw, ds, d_eval, eval_stride, no_fuse ? nullptr : &sink);
    if (!fused) {
        TreeArgs a{};
        a.cols = d_eval;
        a.col_stride = eval_stride;
        a.skip_a = d_layers != nullptr && n >= 1;  // the prover never reads the leaf hashes (plan_merkle_decommit, prover.cpp)
        build_tree(L, T_LEAF4, a, n, d_layers, d_scratch, d_root, tr, tr_init, tr_init_pitch);
        return;
    }
    TreeArgs a{};
    a.tree_log = n;
    a.layers = d_layers;
    a.store_all = d_layers != nullptr;
    a.bstride = L.bstride;
    const uint32_t cur = n - (fused - 1);
    finish_tree(L, a, n, cur, d_layers ? d_layers + merkle_layer_offset(n, cur) : s0, s0, s1, d_root, tr, tr_init, tr_init_pitch);
}

// The fused small-domain kernel needs more LDS than a 64 KB part has: its dynamic block (blob + spare word + coefficients) is
// (15 << L) / 4 + 4 + (4 << L) words = 63.5 KB at L = SMALL_MAX_LOG_COEF, on top of ~31 KB of static arrays (V, RA, QQ): ~94 KB per
// workgroup, one workgroup per CU of a 160 KB gfx950 CU.  Called once per context at creation (the attribute belongs to the current
// device's function object): false = this device cannot run it, the caller switches the context to the general path.
bool small_first_opt_in() {
    int dev = 0, lds_max = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    const size_t dyn = ((((size_t)15 << SMALL_MAX_LOG_COEF) + 3) / 4 + 4 + ((size_t)4 << SMALL_MAX_LOG_COEF)) * sizeof(uint32_t);
    hipFuncAttributes fa{};
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(small_first_kernel)) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (fa.sharedSizeBytes + dyn > (size_t)lds_max) return false;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(small_first_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return true;
}

bool small_domain_shape(const Tuning& tn, uint32_t Lc, uint32_t n) {
    return !tn.no_small_fused /* A/B knob: the general path for every size */ && n >= SMALL_MIN_LOG_DOMAIN && n <= SMALL_MAX_LOG_DOMAIN && Lc <= SMALL_MAX_LOG_COEF && Lc <= n;
}

void small_encode_and_first_tree(const Launch& L, const uint8_t* d_data, size_t len, size_t data_stride, uint32_t Lc, uint32_t n,
                                 const uint32_t* d_tw, DomainScalars ds, uint32_t* d_eval, size_t eval_stride, uint8_t* d_layers,
                                 uint8_t* d_scratch, uint8_t* d_root, DevTranscript* tr, const DevTranscript* tr_init, size_t tr_init_pitch) {
    uint8_t* s0 = d_scratch;
    uint8_t* s1 = d_scratch ? d_scratch + ((size_t)32 << (n > 4 ? n - 4 : 0)) : nullptr;
    SmallFirstArgs a{};
    a.data = d_data;
    a.len = len;
    a.data_stride = data_stride;
    a.L = Lc;
    a.n = n;
    a.init_y = ds.init_y;
    a.tw = d_tw;
    a.eval = d_eval;
    a.eval_stride = eval_stride;
    a.tree.tree_log = n;
    a.tree.level_a = n;
    a.tree.layers = d_layers;
    a.tree.store_all = d_layers != nullptr;
    a.tree.skip_a = d_layers != nullptr;  // the prover never reads the leaf hashes (plan_merkle_decommit, prover.cpp)
    a.tree.last_out = s0;
    a.tree.bstride = L.bstride;
    const double N = (double)((size_t)1 << n);
    {
        // algorithmic bytes: the blob + the encode (16 N (1 + 2^-B)) + leaves (16 B in, 32 B out) + 8 node levels
        Scope scope(L, "small_first", (double)len + 4.0 * 4.0 * (N + (double)((size_t)1 << Lc)) + 48.0 * N + node_levels_bytes(n - 1, T9_LEVELS - 1));
        const size_t lds_words = (((size_t)15 << Lc) + 3) / 4 + 4 + ((size_t)4 << Lc);  // blob + spare word, coefficients: <= 63.5 KB (+ ~31 KB static: small_first_opt_in)
        small_first_kernel<<<dim3(1u << (n - 8), L.batch), 256, lds_words * sizeof(uint32_t), L.stream>>>(a);
    }
    const uint32_t cur = n - (T9_LEVELS - 1);
    finish_tree(L, a.tree, n, cur, d_layers ? d_layers + merkle_layer_offset(n, cur) : s0, s0, s1, d_root, tr, tr_init, tr_init_pitch);
}

void merkle_tree4(const Launch& L, const uint32_t* c0, const uint32_t* c1, const uint32_t* c2, const uint32_t* c3, uint32_t m,
                  uint8_t* d_layers) {
    // columns must be equally strided for the fused kernel; the Level B entry point passes d_cols[4][2^m]
    TreeArgs a{};
    a.cols = c0;
    a.col_stride = (size_t)(c1 - c0);
    (void)c2;
    (void)c3;
    build_tree(L, T_LEAF4, a, m, d_layers, nullptr, nullptr, nullptr);
}

size_t merkle_root_scratch_bytes(uint32_t m) { return ((size_t)64 << (m > 4 ? m - 4 : 0)) + 256; }

void merkle_root4(const Launch& L, const uint32_t* c0, const uint32_t* c1, const uint32_t* c2, const uint32_t* c3, uint32_t m,
                  uint8_t* d_scratch, uint8_t* d_root) {
    TreeArgs a{};
    a.cols = c0;
    a.col_stride = (size_t)(c1 - c0);
    (void)c2;
    (void)c3;
    build_tree(L, T_LEAF4, a, m, nullptr, d_scratch, d_root, nullptr);
}

void tree_first_layer(const Launch& L, const uint32_t* cols, size_t stride, uint32_t m, uint8_t* layers, DevTranscript* tr,
                      const DevTranscript* tr_init, size_t tr_init_pitch) {
    TreeArgs a{};
    a.cols = cols;
    a.col_stride = stride;
    a.skip_a = m >= 1;  // the prover never reads the leaf hashes (plan_merkle_decommit, prover.cpp)
    build_tree(L, T_LEAF4, a, m, layers, nullptr, nullptr, tr, tr_init, tr_init_pitch);
}

void fold_and_tree(const Launch& L, bool circle, const uint32_t* src, size_t src_stride, uint32_t src_log, uint32_t n,
                   const uint32_t* d_itw, DomainScalars ds, uint32_t* dst_vals, uint8_t* layers, DevTranscript* tr) {
    const uint32_t m = src_log - 1;
    TreeArgs a{};
    a.cols = src;
    a.col_stride = src_stride;
    a.out_vals = dst_vals;
    a.out_stride = (size_t)1 << m;
    a.itw = circle ? d_itw : d_itw + tw_level_offset(n, n - 1 - src_log);
    a.n = n;
    a.inv_init_y = ds.inv_init_y;
    a.tr = tr;
    a.skip_a = m >= 1;
    build_tree(L, circle ? T_FOLD_CIRCLE : T_FOLD_LINE, a, m, layers, nullptr, nullptr, tr);
}

void fri_tail(const Launch& L, const uint32_t* src, size_t src_stride, uint32_t src_log, bool src_is_circle, uint32_t n,
              const uint32_t* d_itw, DomainScalars ds, uint32_t last_log, uint32_t last, uint32_t n_layers, uint32_t* const* vals,
              uint8_t* const* trees, DevTranscript* tr, uint32_t* d_gnext) {
    TailArgs a{};
    a.gnext = d_gnext;
    a.src = src;
    a.src_stride = src_stride;
    a.src_log = src_log;
    a.src_is_circle = src_is_circle ? 1 : 0;
    a.n = n;
    a.itw = d_itw;
    a.inv_init_y = ds.inv_init_y;
    a.last_log = last_log;
    a.last = last;
    a.n_layers = n_layers;
    for (uint32_t i = 0; i < n_layers && i < TAIL_MAX_LAYERS; i++) {
        a.vals[i] = vals[i];
        a.trees[i] = trees[i];
    }
    a.tr = tr;
    a.bstride = L.bstride;
    double bytes = 0;
    for (uint32_t i = 0; i < n_layers; i++) bytes += 168.0 * (double)((size_t)1 << (src_log - 1 - i));
    Scope scope(L, "fri_tail", bytes);
    tail_kernel<<<dim3(1, L.batch), WG1_THREADS, 0, L.stream>>>(a);
}

void grind_dev(const Launch& L, DevTranscript* tr, uint32_t* d_next, uint32_t pow_bits, uint64_t base, uint64_t count, bool next_zeroed) {
    GrindArgs a{};
    a.tr = tr;
    a.next = d_next;
    a.pow_bits = pow_bits;
    a.batch = L.batch;
    a.base = base;
    const uint64_t nwin = (count + GRIND_WINDOW - 1) / GRIND_WINDOW;
    a.n_windows = nwin > 0xFFFF0000ull ? 0xFFFF0000u : (uint32_t)nwin;
    if (a.n_windows == 0) return;
    Scope scope(L, "grind", 0.0);
    if (!next_zeroed) (void)hipMemsetAsync(d_next, 0, sizeof(uint32_t) * GRIND_NEXT_STRIDE * L.batch, L.stream);
    // nonces per lane and claim: 4 for a lone blob (the windows in flight when the hit arrives are wasted work: latency) and for big
    // batches (the counters' lines are then spread enough), 8 in between — measured with the counters on their own cache lines:
    // 10 blobs of 2^24: 772 (round 4) -> 397 (4) / 337 (8) / 408 (16) us for 12.9 M nonces (279 us at the compression ceiling);
    // 32 blobs of 2^20: 1089 -> 637 / 702 / 714 us for 25.2 M nonces (548 us)
    const uint32_t iters = L.tune->grind_iters ? L.tune->grind_iters : ((L.batch >= 2 && L.batch < 16) ? 8u : 4u);
    a.units = iters < 4 ? 1u : iters / 4;
    // workgroups in flight: the chip holds 2048 (8 per CU); a lone blob gets no more than cover about half the expected search
    // (2^pow_bits nonces), so that the windows in flight when the first hit arrives are not mostly beyond it
    uint64_t want = (((uint64_t)1 << (pow_bits > 40 ? 40 : pow_bits)) / 2 / GRIND_WINDOW) * L.batch;
    if (want < 64) want = 64;
    if (want > 2048) want = 2048;
    if (want > (uint64_t)a.n_windows * L.batch) want = (uint64_t)a.n_windows * L.batch;
    grind_dev_kernel<<<(unsigned)want, 256, 0, L.stream>>>(a);
}

}  // namespace k
}  // namespace frieda
