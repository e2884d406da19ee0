// reconstruct.cpp — host side of the reconstruction entry points (declarations: include/frieda_hip.h): the blob or the
// coefficient columns back from scattered cells (dense solve: the small system on the host up to 256 cells, blocked Gauss-Jordan on the
// device up to 4096) and from any >= 2^L + 2 sampled points (erasure locator, erasure.hip; product tree from 2^15 coefficients on).
// No exception leaves this file (FR_GUARD_*, host.h).
#include <string.h>

#include <algorithm>
#include <new>

#include "host.h"

using namespace frieda;

extern "C" {

// ---- reconstruction from scattered cells ----
namespace {
// inverse of V[r][u] = prod over the set bits b of u of s_b(c_r), s_b(c) = +- T_{m+b-1}[c >> (b+1)] (minus when bit b of c is set),
// T_l[h] = x-coordinate of C_l.at(brev(h, n-2-l)), C_l = half_odds(n-1) doubled l times (oracle: fo_reconstruct_cells).
// Gauss-Jordan over M31 on the host: R <= 256.  false: singular matrix (repeated cells; no singular set of distinct cells has been
// observed — the code is MDS up to one dimension — but the solve reports it rather than assuming it away).
// row of the cell matrix for cell c: row[u] = prod over the set bits b of u of s_b(c), u < R = 2^nb
void cells_matrix_row(uint32_t c, uint32_t nb, uint32_t m, uint32_t n, uint32_t* row) {
    row[0] = 1;
    for (uint32_t b = 0; b < nb; b++) {
        uint32_t t;
        if (m + b == 0) {
            // single points (m == 0): bit 0 of the index is the circle layer, twiddle Y[c >> 1] = [y, -y, -x, x][h & 3] of the
            // pair (x, y) = (T_0[2 (h >> 2)], T_0[2 (h >> 2) + 1])
            const uint32_t h = c >> 1;
            const Coset cs = Coset::half_odds(n - 1);
            if (n < 3) {
                const uint32_t y = point_from_index(cs.initial).y;
                t = (h & 1u) ? m31_neg(y) : y;
            } else {
                const uint32_t jj = h >> 2, rr = h & 3u;
                const uint32_t v = cs.at(bit_reverse(2 * jj + (rr < 2 ? 1 : 0), n - 2)).x;
                t = (rr == 1 || rr == 2) ? m31_neg(v) : v;
            }
        } else {
            const uint32_t lv = m + b - 1;
            Coset cs = Coset::half_odds(n - 1);
            for (uint32_t i = 0; i < lv; i++) cs = cs.doubled();
            t = cs.at(bit_reverse(c >> (b + 1), n - 2 - lv)).x;
        }
        if ((c >> b) & 1u) t = m31_neg(t);
        for (uint32_t u = 0; u < (1u << b); u++) row[(1u << b) + u] = m31_mul(row[u], t);
    }
}

// Over-determined form: n_avail >= R cells are offered; picks, in the order given, R of them whose rows are linearly independent
// (Gaussian elimination with the pivot taken from the first unused row that has a non-zero entry in the column).  false: the offered
// cells do not span the R-dimensional space.  chosen[k] = position in the caller's list of the k-th cell taken.
bool cells_select_independent(const uint32_t* cell_index, uint32_t n_avail, uint32_t R, uint32_t m, uint32_t n, std::vector<uint32_t>& chosen) {
    uint32_t nb = 0;
    while ((1u << nb) < R) nb++;
    std::vector<uint32_t> A((size_t)n_avail * R);
    for (uint32_t r = 0; r < n_avail; r++) cells_matrix_row(cell_index[r], nb, m, n, &A[(size_t)r * R]);
    std::vector<uint8_t> used(n_avail, 0);
    chosen.clear();
    for (uint32_t col = 0; col < R; col++) {
        uint32_t piv = 0;
        while (piv < n_avail && (used[piv] || A[(size_t)piv * R + col] == 0)) piv++;
        if (piv == n_avail) return false;
        used[piv] = 1;
        chosen.push_back(piv);
        const uint32_t* prow = &A[(size_t)piv * R];
        const uint32_t inv = m31_inv(prow[col]);
        for (uint32_t r = 0; r < n_avail; r++) {
            if (used[r]) continue;
            uint32_t* row = &A[(size_t)r * R];
            if (!row[col]) continue;
            const uint32_t f = m31_mul(row[col], inv);
            for (uint32_t j = col; j < R; j++) row[j] = m31_sub(row[j], m31_mul(f, prow[j]));
        }
    }
    return true;
}

bool cells_matrix_inverse(const uint32_t* cell_index, uint32_t R, uint32_t m, uint32_t n, std::vector<uint32_t>& vinv) {
    std::vector<uint32_t> A((size_t)R * 2 * R, 0u);
    uint32_t nb = 0;
    while ((1u << nb) < R) nb++;
    for (uint32_t r = 0; r < R; r++) {
        uint32_t* row = &A[(size_t)r * 2 * R];
        cells_matrix_row(cell_index[r], nb, m, n, row);
        row[R + r] = 1;
    }
    for (uint32_t col = 0; col < R; col++) {
        uint32_t piv = col;
        while (piv < R && A[(size_t)piv * 2 * R + col] == 0) piv++;
        if (piv == R) return false;
        if (piv != col)
            for (uint32_t j = 0; j < 2 * R; j++) std::swap(A[(size_t)piv * 2 * R + j], A[(size_t)col * 2 * R + j]);
        uint32_t* prow = &A[(size_t)col * 2 * R];
        const uint32_t inv = m31_inv(prow[col]);
        for (uint32_t j = col; j < 2 * R; j++) prow[j] = m31_mul(prow[j], inv);
        for (uint32_t r = 0; r < R; r++) {
            if (r == col) continue;
            uint32_t* row = &A[(size_t)r * 2 * R];
            const uint32_t f = row[col];
            if (!f) continue;
            for (uint32_t j = col; j < 2 * R; j++) row[j] = m31_sub(row[j], m31_mul(f, prow[j]));
        }
    }
    vinv.resize((size_t)R * R);
    for (uint32_t u = 0; u < R; u++)
        for (uint32_t r = 0; r < R; r++) vinv[(size_t)u * R + r] = A[(size_t)u * 2 * R + R + r];
    return true;
}

// coefficients of `ncols` columns from n_cells scattered cells into d_coef[ncols][2^log_coef], or — d_coef == nullptr — into the
// start of the arena (the caller reserved arena_off bytes there).  Scratch: the arena behind arena_off.
// slot_of (optional): cell r of the system is the slot_of[r]-th cell of the caller's d_cells buffer (the over-determined entry)
int interpolate_cells(frieda_ctx* ctx, const uint32_t* d_cells, const uint32_t* cell_index, uint32_t n_cells, uint32_t ncols, uint32_t log_cell,
                      uint32_t log_coef, uint32_t log_domain, uint32_t* d_coef, size_t arena_off, const uint32_t* slot_of = nullptr) {
    Ctx& c = ctx->c;
    FR_NO_JOB(&c);
    if (log_cell > log_coef || log_coef > log_domain || log_domain < 1 || log_domain > FRIEDA_MAX_LOG_DOMAIN)
        return c.fail(FRIEDA_ERR_ARG, "cells: need log_cell <= log_coef <= log_domain");
    if (log_coef - log_cell > FRIEDA_MAX_LOG_CELLS || n_cells != (1u << (log_coef - log_cell)))
        return c.fail(FRIEDA_ERR_ARG, "cells: n_cells must be 2^(log_coef - log_cell) and at most 2^FRIEDA_MAX_LOG_CELLS "
                                      "(with one spare cell, or two spare points, frieda_circle_interpolate_points has no such bound)");
    for (uint32_t r = 0; r < n_cells; r++)
        if ((uint64_t)cell_index[r] >= ((uint64_t)1 << (log_domain - log_cell))) return c.fail(FRIEDA_ERR_ARG, "cells: cell index out of range");
    {
        std::vector<uint32_t> sorted(cell_index, cell_index + n_cells);
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return c.fail(FRIEDA_ERR_ARG, "cells: cell indices are not distinct");
    }
    const size_t M = (size_t)1 << log_cell, w_words = (size_t)n_cells * ncols * M;
    const size_t w_bytes = (4 * w_words + 255) & ~(size_t)255;
    TwiddleSet ts;
    const bool on_device = n_cells > 256;  // the cubic solve moves to the device beyond what the host does in milliseconds
    std::vector<uint32_t> vinv;
    if (!on_device && !cells_matrix_inverse(cell_index, n_cells, log_cell, log_domain, vinv))
        return c.fail(FRIEDA_ERR_ARG, "cells: these cells do not determine the polynomial (singular system)");
    const size_t solve_bytes = on_device ? k::cells_inverse_scratch_bytes(n_cells) : ((4 * vinv.size() + 255) & ~(size_t)255);
    int rc = c.ensure_arena(arena_off + w_bytes + solve_bytes + 512);
    if (rc) return rc;
    rc = c.get_twiddles(log_domain, ts);
    if (rc) return rc;
    uint32_t* d_w = reinterpret_cast<uint32_t*>(c.arena + arena_off);
    uint8_t* d_solve = c.arena + arena_off + w_bytes;
    const uint32_t* d_vinv = reinterpret_cast<const uint32_t*>(d_solve);
    size_t vinv_pitch = n_cells;
    const uint32_t* d_state = nullptr;
    if (on_device) {
        // cell indices behind the solver's own scratch; the blocked Gauss-Jordan leaves V^-1 in the right half of [V | I]
        uint32_t* d_idx = k::cells_inverse_index_buffer(d_solve, n_cells);
        FR_HIP(&c, hipMemcpyAsync(d_idx, cell_index, 4 * (size_t)n_cells, hipMemcpyHostToDevice, c.stream));
        k::cells_matrix_inverse_device(c.launch(), d_idx, n_cells, log_coef - log_cell, log_cell, log_domain, ts.d_tw, d_solve, &d_vinv, &vinv_pitch,
                                       &d_state);
    } else {
        FR_HIP(&c, hipMemcpyAsync(d_solve, vinv.data(), 4 * vinv.size(), hipMemcpyHostToDevice, c.stream));
    }
    for (uint32_t r = 0; r < n_cells; r++)  // undo the block transform of every cell (layers log_cell-1 .. 0 with the cell's twiddles)
        k::circle_interpolate_block(c.launch(), d_cells + (size_t)(slot_of ? slot_of[r] : r) * ncols * M, M, ncols, log_cell, log_domain,
                                    cell_index[r], ts.d_itw, ts.ds, d_w + (size_t)r * ncols * M, M);
    k::cells_combine(c.launch(), d_w, d_vinv, n_cells, ncols, log_cell, d_coef ? d_coef : reinterpret_cast<uint32_t*>(c.arena),
                     (size_t)1 << log_coef, vinv_pitch);
    uint32_t singular = 0;
    if (d_state) FR_HIP(&c, hipMemcpyAsync(&singular, d_state, 4, hipMemcpyDeviceToHost, c.stream));
    FR_HIP(&c, hipStreamSynchronize(c.stream));  // vinv / cell_index are host memory of this call
    FR_HIP(&c, hipGetLastError());
    if (singular) return c.fail(FRIEDA_ERR_ARG, "cells: these cells do not determine the polynomial (singular system)");
    return FRIEDA_OK;
}
}  // namespace

int frieda_circle_interpolate_cells(frieda_ctx* ctx, const uint32_t* d_cells, const uint32_t* cell_index, uint32_t n_cells, uint32_t ncols,
                                    uint32_t log_cell, uint32_t log_coef, uint32_t log_domain, uint32_t* d_coef) {
    if (!ctx || !d_cells || !cell_index || !d_coef || ncols == 0 || ncols > 1024) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    return interpolate_cells(ctx, d_cells, cell_index, n_cells, ncols, log_cell, log_coef, log_domain, d_coef, 0);
    FR_GUARD_END(ctx)
}

int frieda_circle_interpolate_cells_any(frieda_ctx* ctx, const uint32_t* d_cells, const uint32_t* cell_index, uint32_t n_avail, uint32_t ncols,
                                        uint32_t log_cell, uint32_t log_coef, uint32_t log_domain, uint32_t* d_coef, uint32_t* out_used) {
    if (!ctx || !d_cells || !cell_index || !d_coef || ncols == 0 || ncols > 1024) return FRIEDA_ERR_ARG;
    if (log_cell > log_coef || log_coef > log_domain || log_domain < 1 || log_domain > FRIEDA_MAX_LOG_DOMAIN) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    FR_NO_JOB(&ctx->c);  // before the host-side selection below (up to 65536 x 256 x 256 multiplications): a busy context answers at once
    if (log_coef - log_cell > 8) return ctx->c.fail(FRIEDA_ERR_ARG, "cells (any): at most 256 cells are needed in this form (the host-side selection)");
    const uint32_t R = 1u << (log_coef - log_cell);
    if (n_avail < R || n_avail > 65536) return ctx->c.fail(FRIEDA_ERR_ARG, "cells (any): need between 2^(log_coef - log_cell) and 65536 cells");
    for (uint32_t r = 0; r < n_avail; r++)
        if ((uint64_t)cell_index[r] >= ((uint64_t)1 << (log_domain - log_cell))) return ctx->c.fail(FRIEDA_ERR_ARG, "cells: cell index out of range");
    std::vector<uint32_t> chosen;
    if (!cells_select_independent(cell_index, n_avail, R, log_cell, log_domain, chosen))
        return ctx->c.fail(FRIEDA_ERR_ARG, "cells (any): the offered cells do not determine the polynomial");
    std::vector<uint32_t> idx(R);
    for (uint32_t k2 = 0; k2 < R; k2++) idx[k2] = cell_index[chosen[k2]];
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    const int rc = interpolate_cells(ctx, d_cells, idx.data(), R, ncols, log_cell, log_coef, log_domain, d_coef, 0, chosen.data());
    if (rc == FRIEDA_OK && out_used) memcpy(out_used, chosen.data(), 4 * (size_t)R);
    return rc;
    FR_GUARD_END(ctx)
}

int frieda_reconstruct_cells_device(frieda_ctx* ctx, const uint32_t* d_cells, const uint32_t* cell_index, uint32_t n_cells, uint32_t log_cell,
                                    uint32_t log_coef, uint32_t log_domain, size_t len, void* d_out_bytes) {
    if (!ctx || !d_cells || !cell_index || (len && !d_out_bytes)) return FRIEDA_ERR_ARG;
    if (log_coef > FRIEDA_MAX_LOG_DOMAIN) return FRIEDA_ERR_ARG;
    const size_t n_felts = (size_t)4 << log_coef;
    FR_GUARD_BEGIN
    if ((8 * len + 29) / 30 > n_felts) return ctx->c.fail(FRIEDA_ERR_ARG, "len does not fit the polynomial");
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    const size_t coef_bytes = (sizeof(uint32_t) * n_felts + 255) & ~(size_t)255;
    int rc = interpolate_cells(ctx, d_cells, cell_index, n_cells, 4, log_cell, log_coef, log_domain, nullptr, coef_bytes);
    if (rc) return rc;
    k::pack30(ctx->c.launch(), reinterpret_cast<const uint32_t*>(ctx->c.arena), n_felts, static_cast<uint8_t*>(d_out_bytes), len);
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

// ---- reconstruction from any >= 2^log_coef + 2 points (erasure.hip) ----
namespace {
// coefficients of `ncols` columns into d_coef[ncols][2^log_coef], or — d_coef == nullptr — into the start of the arena (arena_off bytes
// reserved there by the caller).  Cells as in interpolate_cells (runs of 2^log_cell entries; repeated cells are dropped).
int interpolate_points(frieda_ctx* ctx, const uint32_t* d_cells, const uint32_t* cell_index, uint32_t n_cells, uint32_t ncols, uint32_t log_cell,
                       uint32_t log_coef, uint32_t log_domain, uint32_t* d_coef, size_t arena_off) {
    Ctx& c = ctx->c;
    FR_NO_JOB(&c);
    if (log_cell > log_domain || log_coef > log_domain || log_coef < 1 || log_domain < 2 || log_domain + 1 > FRIEDA_MAX_LOG_DOMAIN)
        return c.fail(FRIEDA_ERR_ARG, "points: need 1 <= log_coef <= log_domain <= FRIEDA_MAX_LOG_DOMAIN - 1 and log_cell <= log_domain");
    const size_t N = (size_t)1 << log_domain, K = (size_t)1 << log_coef;
    const uint32_t n = log_domain;
    // The distinct sampled positions (first occurrence of every cell wins) and where their values sit in the caller's buffer are worked out
    // on the device (erasure_sample_lists): only their number comes back, for the argument checks below.
    const uint32_t domain_cells = (uint32_t)(N >> log_cell);
    // n_cells * ncols * 2^log_cell <= 2^32, checked without forming the product (n_cells < 2^32, ncols <= 2^10, log_cell <= 27: it can reach
    // 2^69); every 32-bit position computed in erasure.hip relies on this bound
    if ((uint64_t)n_cells > (0x100000000ull >> log_cell) / ncols) return c.fail(FRIEDA_ERR_ARG, "points: sample buffer beyond 2^32 words");
    const size_t s_cap = (size_t)n_cells << log_cell;  // points offered, repeats included
    if (s_cap < K + 2)
        return c.fail(FRIEDA_ERR_ARG, "points: need at least 2^log_coef + 2 distinct points (the locator polynomial needs two spare samples)");
    // S, the points the locator is built from (all offered points serve the check): the first K + 2 single points, Z_S a product of
    // lines through pairs — or, for samples in cells of M >= 2 entries, the first K / M + 1 whole cells, Z_S a product over cells.
    // Single points of a large polynomial take Z_S through a product tree (O(K log^2 K)) instead of K / 2 lines at each of K points
    // (O(K^2)); many small cells do too, as the single points they consist of (the per-cell form costs K * K / M factor evaluations:
    // 38 ms against 4 ms for 2^16 cells of 16 on a 2^24 domain, 3.1 against 1.6 ms for 2^14 of them at 2^22; cells of 256 are level
    // there).  FRIEDA_ERASURE_TREE_MIN_LOG: smallest log_coef that takes the tree (default 15: 0.67 against 0.97 ms there; the parity
    // tests lower it per context; 32 = never).
    const uint32_t tree_min_log = c.tuning.erasure_tree_min_log;
    const bool by_cells = log_cell >= 1 && !(log_coef >= tree_min_log && 2 * log_coef >= 32 + log_cell);
    const uint32_t n_use_cells = by_cells ? (uint32_t)(K >> log_cell) + 1 : 0;
    const uint32_t s_use = by_cells ? (uint32_t)((size_t)n_use_cells << log_cell) : (uint32_t)K + 2;
    const uint32_t n_lines = by_cells ? n_use_cells : s_use / 2;  // factors of Z_S
    const bool by_tree = !by_cells && log_coef >= tree_min_log;
    if (s_use > s_cap) return c.fail(FRIEDA_ERR_ARG, "points: cells of 2^log_cell entries: need 2^(log_coef - log_cell) + 1 distinct cells");

    // domains: D (log n) and the next canonic domain D' (log n + 1)
    auto make_domain = [](uint32_t lg) {
        k::ErasureDomain g;
        memset(&g, 0, sizeof g);
        const Coset h = Coset::half_odds(lg - 1);
        g.init = point_from_index(h.initial);
        CPoint sp = point_from_index(h.step);
        for (uint32_t b = 0; b + 1 < lg && b < 32; b++) {
            g.step_pow[b] = sp;
            sp = cp_double(sp);
        }
        g.n = lg;
        return g;
    };
    const k::ErasureDomain g0 = make_domain(n), g1 = make_domain(n + 1);

    // workspace
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t chunks = 64;  // upper bound of erasure_zpart_chunks
    ArenaPlan plan;
    plan.off = arena_off;
    const size_t o_pos = plan.take(4 * s_cap), o_src = plan.take(4 * s_cap);
    const size_t ded_chunks = k::erasure_sample_lists_chunks(n_cells);
    const size_t o_idx = plan.take(4 * (size_t)n_cells), o_fc = plan.take(4 * (size_t)n_cells), o_fr = plan.take(4 * (size_t)n_cells);
    const size_t o_own = plan.take(4 * (size_t)domain_cells), o_csum = plan.take(4 * ded_chunks), o_coff = plan.take(4 * ded_chunks), o_state = plan.take(8);
    const size_t o_la = plan.take(4 * (size_t)n_lines), o_lb = plan.take(4 * (size_t)n_lines), o_lc = plan.take(4 * (size_t)n_lines);
    const size_t s_max = std::max<size_t>(s_use, K);
    const size_t o_px = plan.take(4 * s_max), o_py = plan.take(4 * s_max);
    const size_t o_zp = plan.take(by_tree ? 0 : 4 * chunks * s_max), o_z = plan.take(4 * s_max), o_bad = plan.take(4);  // (o_zp: partial products of the direct routes)
    const size_t o_w = plan.take(al(4 * N) * ncols);   // Z * p on D; later the re-encoded polynomial for the check
    const size_t o_q = plan.take(al(4 * N) * ncols);   // coefficients of Z * p
    const size_t o_ev = plan.take(al(8 * N) * ncols);  // Z * p on D'
    const size_t o_blk = plan.take(al(4 * K) * ncols);
    const size_t o_ta = plan.take(by_tree ? 8 * K : 0), o_tb = plan.take(by_tree ? 8 * K : 0), o_tc = plan.take(by_tree ? 16 * K : 0);
    int rc = c.ensure_arena(plan.off);
    if (rc) return rc;
    TwiddleSet ts0, ts1;
    rc = c.get_twiddles(n, ts0);
    if (rc) return rc;
    rc = c.get_twiddles(n + 1, ts1);
    if (rc) return rc;
    uint8_t* A = c.arena;
    auto W32 = [&](size_t off) { return reinterpret_cast<uint32_t*>(A + off); };
    hipStream_t s = c.stream;
    const k::Launch LN = c.launch();
    FR_HIP(&c, hipMemcpyAsync(A + o_idx, cell_index, 4 * (size_t)n_cells, hipMemcpyHostToDevice, s));
    FR_HIP(&c, k::erasure_sample_lists(LN, W32(o_idx), n_cells, domain_cells, ncols, log_cell, W32(o_own), W32(o_csum), W32(o_coff), W32(o_fc), W32(o_fr),
                            W32(o_state), W32(o_pos), W32(o_src)));
    uint32_t state[2] = {0, 0};
    FR_HIP(&c, hipMemcpyAsync(state, A + o_state, 8, hipMemcpyDeviceToHost, s));
    FR_HIP(&c, hipMemsetAsync(A + o_bad, 0, 4, s));
    FR_HIP(&c, hipStreamSynchronize(s));  // (cell_index is host memory of this call; the counts decide whether there is anything to do)
    FR_HIP(&c, hipGetLastError());
    if (state[1]) return c.fail(FRIEDA_ERR_ARG, "points: cell index out of range");
    const uint32_t s_all = (uint32_t)((size_t)state[0] << log_cell);  // distinct points offered
    if (s_all < K + 2)
        return c.fail(FRIEDA_ERR_ARG, "points: need at least 2^log_coef + 2 distinct points (the locator polynomial needs two spare samples)");
    if (s_use > s_all) return c.fail(FRIEDA_ERR_ARG, "points: cells of 2^log_cell entries: need 2^(log_coef - log_cell) + 1 distinct cells");
    const size_t w_stride = al(4 * N) / 4, ev_stride = al(8 * N) / 4, blk_stride = al(4 * K) / 4;
    uint32_t* coef_out = d_coef ? d_coef : reinterpret_cast<uint32_t*>(A);
    const uint32_t* ev_first = nullptr;  // Z * p on the first 2^log_coef entries of D' (step 3)
    size_t ev_first_stride = 0;
    // 1. the locator on the points it is built from: ratio of tangent derivatives (direct routes) or V_D / Z_S carried to D (tree)
    if (!by_tree) k::erasure_points(LN, g0, W32(o_pos), s_use, W32(o_px), W32(o_py));
    if (by_cells) {
        k::erasure_cell_firsts(LN, W32(o_pos), n_use_cells, log_cell, W32(o_lb));  // (o_lb: free in this form)
        k::erasure_cellconst(LN, g0, W32(o_lb), n_use_cells, log_cell, W32(o_la));
        k::erasure_zeval_cells(LN, W32(o_px), s_use, log_cell, W32(o_la), n_use_cells, true, W32(o_zp), W32(o_z));
        k::erasure_known_weights_cells(LN, W32(o_px), s_use, n, log_cell, W32(o_z));
    } else if (!by_tree) {
        k::erasure_lines(LN, g0, W32(o_pos), n_lines, W32(o_la), W32(o_lb), W32(o_lc));
        k::erasure_zeval(LN, W32(o_px), W32(o_py), s_use, W32(o_la), W32(o_lb), W32(o_lc), n_lines, true, W32(o_zp), W32(o_z));
        k::erasure_known_weights(LN, W32(o_px), W32(o_py), s_use, n, W32(o_z));
    } else {
        // Z_S by a tree over leaves of 32 lines (the one line beyond K / 2 rides in leaf 0).  A node over 2^j * 32 lines (degree 2^j * 32) is held as its
        // values on the canonic domain of 2^(j + 7) points; two children go to the parent's domain through their coefficients (the canonic
        // domains of different sizes share no points) and multiply pointwise there.  The root (degree K / 2 + 1, 2 K values) -> coefficients ->
        // all of D' (o_ev, free until step 3).  Then Z_E = V_D / Z_S: its values on the first N points of D' -> its N coefficients -> its
        // values on D, of which the S entries are the weights of step 2 (no derivative needed on this route).
        k::erasure_lines(LN, g0, W32(o_pos), n_lines, W32(o_la), W32(o_lb), W32(o_lc));
        const k::ErasureDomain g7 = make_domain(7);
        k::erasure_points(LN, g7, nullptr, 128, W32(o_px), W32(o_py));
        uint32_t nodes = (uint32_t)(K / 64), d = 7;  // K / 2 lines in leaves of 32
        k::erasure_lines32(LN, W32(o_px), W32(o_py), W32(o_la), W32(o_lb), W32(o_lc), n_lines, nodes, W32(o_ta));
        const uint32_t col_chunk = 32768;
        while (nodes > 1) {
            TwiddleSet tsd, tse;
            rc = c.get_twiddles(d, tsd);
            if (rc) return rc;
            rc = c.get_twiddles(d + 1, tse);
            if (rc) return rc;
            const size_t sz = (size_t)1 << d;
            for (uint32_t at = 0; at < nodes; at += col_chunk) {
                const uint32_t cnt = std::min(col_chunk, nodes - at);
                k::circle_interpolate_block(LN, W32(o_ta) + at * sz, sz, cnt, d, d, 0, tsd.d_itw, tsd.ds, W32(o_tb) + at * sz, sz);
                k::circle_evaluate(LN, W32(o_tb) + at * sz, sz, cnt, d, d + 1, tse.d_tw, tse.ds, W32(o_tc) + (size_t)at * 2 * sz, 2 * sz);
            }
            k::erasure_pairmul(LN, W32(o_tc), nodes, (uint32_t)(2 * sz), W32(o_ta));
            nodes /= 2;
            d++;
        }
        // d == log_coef + 1 here: the root's 2 K values -> coefficients -> D' (2 N points)
        TwiddleSet tsr;
        rc = c.get_twiddles(d, tsr);
        if (rc) return rc;
        k::circle_interpolate_block(LN, W32(o_ta), (size_t)1 << d, 1, d, d, 0, tsr.d_itw, tsr.ds, W32(o_tb), (size_t)1 << d);
        k::circle_evaluate_prefix(LN, W32(o_tb), (size_t)1 << d, 1, d, n + 1, n, ts1.d_tw, ts1.ds, W32(o_ev), ev_stride);  // (first half of D': all that is read)
        FR_HIP(&c, hipMemcpyAsync(A + o_tc, A + o_ev, 4 * K, hipMemcpyDeviceToDevice, s));  // Z_S on the block step 3 divides on
        k::erasure_ze(LN, g1, W32(o_ev), (uint32_t)N, n, W32(o_q));
        k::circle_interpolate_block(LN, W32(o_q), w_stride, 1, n, n + 1, 0, ts1.d_itw, ts1.ds, W32(o_w), w_stride);
        k::circle_evaluate(LN, W32(o_w), w_stride, 1, n, n, ts0.d_tw, ts0.ds, W32(o_q), w_stride);
        k::erasure_gather(LN, W32(o_q), W32(o_pos), s_use, W32(o_z));
    }
    // 2. Z * p on D -> its coefficients
    FR_HIP(&c, hipMemsetAsync(A + o_w, 0, al(4 * N) * ncols, s));
    k::erasure_scatter(LN, d_cells, W32(o_src), W32(o_pos), W32(o_z), s_use, ncols, log_cell, W32(o_w), w_stride);
    k::circle_interpolate_block(LN, W32(o_w), w_stride, ncols, n, n, 0, ts0.d_itw, ts0.ds, W32(o_q), w_stride);
    // 3. onto the first 2^log_coef entries of D': p = (Z p) Z_S / V_D there.  Only that block is wanted, so the coefficient vector is
    //    folded down to 2^log_coef entries (four layers per launch, buffers alternating) and a transform of that size finishes.
    {
        const uint32_t* cur = W32(o_q);
        size_t cur_stride = w_stride;
        uint32_t cur_log = n;
        bool into_ev = true;
        while (cur_log > log_coef) {
            const uint32_t cnt = std::min<uint32_t>(4, cur_log - log_coef), out_log = cur_log - cnt;
            uint32_t* dst = into_ev ? W32(o_ev) : W32(o_w);
            const size_t dst_stride = into_ev ? ev_stride : w_stride;
            k::erasure_fold_prefix(LN, cur, cur_stride, ncols, out_log, cnt, ts1.d_tw, n + 1, dst, dst_stride);
            cur = dst;
            cur_stride = dst_stride;
            cur_log = out_log;
            into_ev = !into_ev;
        }
        // (the transform's output takes the buffer the last fold did not write; step 2's o_q is free again)
        uint32_t* dst = cur == W32(o_ev) ? W32(o_q) : W32(o_ev);
        const size_t dst_stride = cur == W32(o_ev) ? w_stride : ev_stride;
        k::circle_evaluate_prefix(LN, cur, cur_stride, ncols, log_coef, n + 1, log_coef, ts1.d_tw, ts1.ds, dst, dst_stride);
        ev_first = dst;
        ev_first_stride = dst_stride;
    }
    k::erasure_points(LN, g1, nullptr, (uint32_t)K, W32(o_px), W32(o_py));
    if (by_cells)
        k::erasure_zeval_cells(LN, W32(o_px), (uint32_t)K, log_cell, W32(o_la), n_use_cells, false, W32(o_zp), W32(o_z));
    else if (by_tree)
        FR_HIP(&c, hipMemcpyAsync(A + o_z, A + o_tc, 4 * K, hipMemcpyDeviceToDevice, s));
    else
        k::erasure_zeval(LN, W32(o_px), W32(o_py), (uint32_t)K, W32(o_la), W32(o_lb), W32(o_lc), n_lines, false, W32(o_zp), W32(o_z));
    k::erasure_divide(LN, ev_first, ev_first_stride, W32(o_z), W32(o_px), (uint32_t)K, ncols, n, W32(o_blk), blk_stride);
    // 4. that block back to coefficients
    k::circle_interpolate_block(LN, W32(o_blk), blk_stride, ncols, log_coef, n + 1, 0, ts1.d_itw, ts1.ds, coef_out, K);
    // 5. encode again and compare every sample that was offered
    k::circle_evaluate(LN, coef_out, K, ncols, log_coef, n, ts0.d_tw, ts0.ds, W32(o_w), w_stride);
    k::erasure_check(LN, d_cells, W32(o_src), W32(o_pos), s_all, ncols, log_cell, W32(o_w), w_stride, W32(o_bad));
    uint32_t bad = 0;
    FR_HIP(&c, hipMemcpyAsync(&bad, A + o_bad, 4, hipMemcpyDeviceToHost, s));
    FR_HIP(&c, hipStreamSynchronize(s));
    FR_HIP(&c, hipGetLastError());
    if (bad)
        return c.fail(FRIEDA_ERR_ARG, "points: the samples are not values of one polynomial of 2^log_coef coefficients (" + std::to_string(bad) +
                                          " sample words differ from the re-encoded result)");
    return FRIEDA_OK;
}
}  // namespace

int frieda_circle_interpolate_points(frieda_ctx* ctx, const uint32_t* d_cells, const uint32_t* cell_index, uint32_t n_cells, uint32_t ncols,
                                     uint32_t log_cell, uint32_t log_coef, uint32_t log_domain, uint32_t* d_coef) {
    if (!ctx || !d_cells || !cell_index || !d_coef || ncols == 0 || ncols > 1024 || n_cells == 0) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    return interpolate_points(ctx, d_cells, cell_index, n_cells, ncols, log_cell, log_coef, log_domain, d_coef, 0);
    FR_GUARD_END(ctx)
}

int frieda_reconstruct_points_device(frieda_ctx* ctx, const uint32_t* d_cells, const uint32_t* cell_index, uint32_t n_cells, uint32_t log_cell,
                                     uint32_t log_coef, uint32_t log_domain, size_t len, void* d_out_bytes) {
    if (!ctx || !d_cells || !cell_index || (len && !d_out_bytes) || n_cells == 0) return FRIEDA_ERR_ARG;
    if (log_coef > FRIEDA_MAX_LOG_DOMAIN) return FRIEDA_ERR_ARG;
    const size_t n_felts = (size_t)4 << log_coef;
    FR_GUARD_BEGIN
    if ((8 * len + 29) / 30 > n_felts) return ctx->c.fail(FRIEDA_ERR_ARG, "len does not fit the polynomial");
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    const size_t coef_bytes = (sizeof(uint32_t) * n_felts + 255) & ~(size_t)255;
    int rc = interpolate_points(ctx, d_cells, cell_index, n_cells, 4, log_cell, log_coef, log_domain, nullptr, coef_bytes);
    if (rc) return rc;
    k::pack30(ctx->c.launch(), reinterpret_cast<const uint32_t*>(ctx->c.arena), n_felts, static_cast<uint8_t*>(d_out_bytes), len);
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

}  // extern "C"
