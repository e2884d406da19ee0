"""GPU: reconstruction from any >= 2^L + 2 sampled points of the codeword, with no bound on their number (erasure.hip,
frieda_circle_interpolate_points / frieda_reconstruct_points_device) — the README's sample() flow (/root/reference/README.md:56-69; not in
/root/reference/src) at the size of the reference's own `blob` fixture.  Parity: the oracle's restatement (fo_reconstruct_points, its own
transforms) and the oracle's dense solve (fo_reconstruct_cells) on small sizes; encode -> sample -> reconstruct == identity at full sizes."""
import ctypes as C

import numpy as np
import pytest

from conftest import splitmix64_bytes
from util import DevBuf

pytestmark = pytest.mark.gpu

P = 2**31 - 1


def _check(ctx, rc):
    from frieda_amd.api import _check as chk

    chk(rc, ctx._h)


def _cells(ev, idx, m):
    return np.ascontiguousarray(np.stack([ev[:, int(c) << m : (int(c) + 1) << m] for c in idx]))  # [R, ncols, 2^m]


@pytest.mark.parametrize(
    "L,n,m,extra,ncols",
    [(1, 2, 0, 2, 1), (1, 3, 0, 2, 4), (2, 4, 0, 2, 4), (3, 6, 0, 3, 2), (4, 8, 0, 5, 4), (5, 7, 0, 2, 4), (6, 10, 0, 40, 3), (8, 12, 0, 2, 4),
     (8, 12, 2, 1, 4), (6, 7, 1, 1, 4), (9, 13, 3, 7, 4), (10, 14, 0, 2, 4), (10, 11, 0, 1022, 4),
     (3, 8, 5, 0, 4), (4, 9, 4, 1, 2), (5, 6, 5, 1, 4), (8, 10, 8, 0, 4), (10, 14, 1, 1, 4), (12, 16, 5, 3, 4)],  # cells: Z_S as a product over cells
)
def test_interpolate_points_vs_oracle(gpu_ctx, oracle, L, n, m, extra, ncols):
    """extra: cells offered beyond the 2^(L - m) that carry 2^L points (m == 0: two spare points are the minimum; negative: all of the
    domain minus something).  Against the oracle's erasure-locator restatement, its dense solve where that applies, and the truth."""
    rng = np.random.default_rng(7000 + 100 * L + 10 * n + m)
    coef = rng.integers(0, P, (ncols, 1 << L), dtype=np.uint32)
    ev = oracle.circle_evaluate(coef, n)
    base = (1 << (L - m)) if m <= L else 0  # whole cells that carry 2^L points (none when one cell already holds more than that)
    n_cells = base + extra
    if m > 0:
        n_cells = max(n_cells, base + 1)  # one more cell of >= 2 points carries the two spare samples
    n_cells = min(n_cells, 1 << (n - m))
    idx = rng.permutation(1 << (n - m))[:n_cells].astype(np.uint32)
    cells = _cells(ev, idx, m)
    pos = (idx[:, None].astype(np.uint64) * (1 << m) + np.arange(1 << m, dtype=np.uint64)[None, :]).ravel().astype(np.uint32)
    vals = np.ascontiguousarray(cells.transpose(0, 2, 1).reshape(-1, ncols))  # [n_pts, ncols] in the order of pos
    if n <= 12:
        assert np.array_equal(oracle.reconstruct_points(vals, pos, n, L), coef)
    d_cells, d_c = DevBuf.from_array(gpu_ctx, cells), DevBuf(gpu_ctx, 4 * ncols << L)
    _check(gpu_ctx, gpu_ctx._L.frieda_circle_interpolate_points(gpu_ctx._h, d_cells.ptr, idx.ctypes.data, n_cells, ncols, m, L, n, d_c.ptr))
    assert np.array_equal(d_c.to_array(np.uint32, (ncols, 1 << L)), coef)
    # repeated cells are ignored (first occurrence wins)
    rep = min(3, n_cells)
    idx2 = np.concatenate([idx, idx[:rep]])
    cells2 = np.concatenate([cells, np.zeros_like(cells[:rep])])
    d_cells2 = DevBuf.from_array(gpu_ctx, cells2)
    _check(gpu_ctx, gpu_ctx._L.frieda_circle_interpolate_points(gpu_ctx._h, d_cells2.ptr, idx2.ctypes.data, n_cells + rep, ncols, m, L, n, d_c.ptr))
    assert np.array_equal(d_c.to_array(np.uint32, (ncols, 1 << L)), coef)


def test_interpolate_points_argument_errors(gpu_ctx, oracle):
    rng = np.random.default_rng(3)
    L, n = 4, 8
    coef = rng.integers(0, P, (4, 1 << L), dtype=np.uint32)
    ev = oracle.circle_evaluate(coef, n)
    idx = rng.permutation(1 << n)[:18].astype(np.uint32)
    d_cells, d_c = DevBuf.from_array(gpu_ctx, _cells(ev, idx, 0)), DevBuf(gpu_ctx, 16 << L)
    f = gpu_ctx._L.frieda_circle_interpolate_points
    h = gpu_ctx._h
    assert f(h, d_cells.ptr, idx.ctypes.data, 17, 4, 0, L, n, d_c.ptr) == 1  # 2^L + 1 points: one short
    assert "2^log_coef + 2" in gpu_ctx._L.frieda_last_error(h).decode()
    dup = idx.copy()
    dup[17] = dup[0]
    assert f(h, d_cells.ptr, dup.ctypes.data, 18, 4, 0, L, n, d_c.ptr) == 1  # 18 offered, 17 distinct
    bad = idx.copy()
    bad[5] = 1 << n
    assert f(h, d_cells.ptr, bad.ctypes.data, 18, 4, 0, L, n, d_c.ptr) == 1  # position outside the domain
    assert f(h, d_cells.ptr, idx.ctypes.data, 18, 4, 0, 0, n, d_c.ptr) == 1  # a constant polynomial: use the cells entry
    assert f(h, d_cells.ptr, idx.ctypes.data, 18, 4, 0, L, 28, d_c.ptr) == 1  # beyond the supported domain (needs a 2^(n+1) transform)
    assert f(h, d_cells.ptr, idx.ctypes.data, 0, 4, 0, L, n, d_c.ptr) == 1
    assert f(h, None, idx.ctypes.data, 18, 4, 0, L, n, d_c.ptr) == 1
    _check(gpu_ctx, f(h, d_cells.ptr, idx.ctypes.data, 18, 4, 0, L, n, d_c.ptr))
    assert np.array_equal(d_c.to_array(np.uint32, (4, 1 << L)), coef)


@pytest.mark.parametrize("L,n,extra,where", [(4, 8, 2, "used"), (6, 10, 30, "used"), (6, 10, 30, "spare"), (10, 14, 100, "spare"), (10, 14, 2, "used")])
def test_inconsistent_samples_are_reported(gpu_ctx, oracle, L, n, extra, where):
    """One corrupted sample word — among the 2^L + 2 points the locator is built from, or among the spare ones that only serve the check —
    is reported (FRIEDA_ERR_ARG), never answered with a wrong polynomial; the oracle's restatement reports the same."""
    rng = np.random.default_rng(8000 + L + n)
    coef = rng.integers(0, P, (4, 1 << L), dtype=np.uint32)
    ev = oracle.circle_evaluate(coef, n)
    n_pts = (1 << L) + extra
    idx = rng.permutation(1 << n)[:n_pts].astype(np.uint32)
    cells = _cells(ev, idx, 0)
    victim = 5 if where == "used" else n_pts - 1
    cells[victim, 2, 0] = (int(cells[victim, 2, 0]) + 1) % P
    d_cells, d_c = DevBuf.from_array(gpu_ctx, cells), DevBuf(gpu_ctx, 16 << L)
    assert gpu_ctx._L.frieda_circle_interpolate_points(gpu_ctx._h, d_cells.ptr, idx.ctypes.data, n_pts, 4, 0, L, n, d_c.ptr) == 1
    assert "not values of one polynomial" in gpu_ctx._L.frieda_last_error(gpu_ctx._h).decode()
    if n <= 10:
        with pytest.raises(ValueError, match="not values of one polynomial"):
            oracle.reconstruct_points(np.ascontiguousarray(cells[:, :, 0]), idx, n, L)
    # the untouched columns alone are fine
    good = np.ascontiguousarray(cells[:, :2, :])
    d_good = DevBuf.from_array(gpu_ctx, good)
    _check(gpu_ctx, gpu_ctx._L.frieda_circle_interpolate_points(gpu_ctx._h, d_good.ptr, idx.ctypes.data, n_pts, 2, 0, L, n, d_c.ptr))
    assert np.array_equal(d_c.to_array(np.uint32, (2, 1 << L)), coef[:2])


@pytest.mark.parametrize("L,n,extra,ncols", [(6, 7, 2, 4), (6, 10, 40, 3), (7, 9, 2, 4), (8, 12, 2, 4), (10, 11, 1022, 4), (10, 14, 2, 4), (12, 13, 2, 4),
                                             (13, 17, 5, 2), (14, 18, 2, 4), (16, 17, 2, 1)])
def test_product_tree_route_matches_line_route(gpu_ctx, oracle, monkeypatch, L, n, extra, ncols):
    """Single points of a polynomial of >= 2^17 coefficients take Z_S through a product tree (O(K log^2 K)) instead of line by line
    (O(K^2)); FRIEDA_ERASURE_TREE_MIN_LOG lowers the threshold so that both routes run on the same small inputs: identical coefficients,
    equal to the truth (and to the oracle's restatement where it is quick), and a corrupted sample is still reported."""
    rng = np.random.default_rng(9100 + 100 * L + n)
    coef = rng.integers(0, P, (ncols, 1 << L), dtype=np.uint32)
    ev = oracle.circle_evaluate(coef, n)
    n_pts = min((1 << L) + extra, 1 << n)
    idx = rng.permutation(1 << n)[:n_pts].astype(np.uint32)
    cells = _cells(ev, idx, 0)
    d_cells, d_c = DevBuf.from_array(gpu_ctx, cells), DevBuf(gpu_ctx, 4 * ncols << L)
    got = {}
    for route, min_log in (("lines", "32"), ("tree", "6")):
        gpu_ctx.set_option("FRIEDA_ERASURE_TREE_MIN_LOG", int(min_log))
        poison = np.full((ncols, 1 << L), 0xEEEEEEEE, dtype=np.uint32)  # the second route must write its own answer
        _check(gpu_ctx, gpu_ctx._L.frieda_dev_upload(gpu_ctx._h, d_c.ptr, poison.ctypes.data, poison.nbytes))
        _check(gpu_ctx, gpu_ctx._L.frieda_circle_interpolate_points(gpu_ctx._h, d_cells.ptr, idx.ctypes.data, n_pts, ncols, 0, L, n, d_c.ptr))
        got[route] = d_c.to_array(np.uint32, (ncols, 1 << L)).copy()
    gpu_ctx.set_option("FRIEDA_ERASURE_TREE_MIN_LOG", 6)  # (the corrupted-sample checks below take the tree route; reset at the end)
    assert np.array_equal(got["tree"], coef) and np.array_equal(got["lines"], coef)
    if n <= 12:
        vals = np.ascontiguousarray(cells[:, :, 0])
        assert np.array_equal(oracle.reconstruct_points(vals, idx, n, L), coef)
    # a corrupted word among the points the locator is built from, tree route
    cells[3, ncols - 1, 0] = (int(cells[3, ncols - 1, 0]) + 1) % P
    d_bad = DevBuf.from_array(gpu_ctx, cells)
    rc_bad = gpu_ctx._L.frieda_circle_interpolate_points(gpu_ctx._h, d_bad.ptr, idx.ctypes.data, n_pts, ncols, 0, L, n, d_c.ptr)
    msg = gpu_ctx._L.frieda_last_error(gpu_ctx._h).decode()
    gpu_ctx.set_option("FRIEDA_ERASURE_TREE_MIN_LOG", 15)  # the default, for the tests that follow on this context
    assert rc_bad == 1 and "not values of one polynomial" in msg


@pytest.mark.parametrize("L,n,m", [(14, 18, 0), (15, 17, 0), (12, 16, 2), (16, 20, 0)])
def test_heavily_repeated_sample_list(gpu_ctx, oracle, L, n, m):
    """The caller's list is de-duplicated on the device (owner table + ranked compaction over 2048-entry chunks): a list in which every
    cell shows up one to four times, shuffled, with garbage in every copy but the first, still gives the truth — the first occurrence
    counts, in list order, across chunk boundaries.  And a list whose distinct cells fall one short is refused."""
    rng = np.random.default_rng(6400 + L + n + m)
    ncols = 4
    coef = rng.integers(0, P, (ncols, 1 << L), dtype=np.uint32)
    d_c0, d_ev = DevBuf.from_array(gpu_ctx, coef), DevBuf(gpu_ctx, 4 * ncols << n)
    _check(gpu_ctx, gpu_ctx._L.frieda_circle_evaluate(gpu_ctx._h, d_c0.ptr, ncols, L, n, d_ev.ptr))
    ev = d_ev.to_array(np.uint32, (ncols, 1 << n))
    need = ((1 << (L - m)) + 1) if m > 0 else (1 << L) + 2
    distinct = rng.permutation(1 << (n - m))[:need].astype(np.uint32)
    copies = rng.integers(1, 5, need)
    lst = np.repeat(distinct, copies)
    order = rng.permutation(lst.size)
    lst = lst[order]
    cells = np.ascontiguousarray(ev.reshape(ncols, -1, 1 << m)[:, lst, :].transpose(1, 0, 2))  # [n_list, ncols, 2^m]
    seen = set()
    for r, ci in enumerate(lst):  # every copy after the first carries garbage
        if int(ci) in seen:
            cells[r] = rng.integers(0, P, cells[r].shape, dtype=np.uint32)
        seen.add(int(ci))
    d_cells, d_c = DevBuf.from_array(gpu_ctx, cells), DevBuf(gpu_ctx, 4 * ncols << L)
    _check(gpu_ctx, gpu_ctx._L.frieda_circle_interpolate_points(gpu_ctx._h, d_cells.ptr, lst.ctypes.data, lst.size, ncols, m, L, n, d_c.ptr))
    assert np.array_equal(d_c.to_array(np.uint32, (ncols, 1 << L)), coef)
    # one distinct cell fewer (all its copies removed): not enough samples, whatever the length of the list
    keep = lst != distinct[0]
    lst2, cells2 = np.ascontiguousarray(lst[keep]), np.ascontiguousarray(cells[keep])
    assert lst2.size >= need - 1
    d_cells2 = DevBuf.from_array(gpu_ctx, cells2)
    assert gpu_ctx._L.frieda_circle_interpolate_points(gpu_ctx._h, d_cells2.ptr, lst2.ctypes.data, lst2.size, ncols, m, L, n, d_c.ptr) == 1
    assert "distinct" in gpu_ctx._L.frieda_last_error(gpu_ctx._h).decode()


def test_random_shapes_both_routes(gpu_ctx, oracle, monkeypatch):
    """Seeded sweep over (coefficients, blow-up, cell size, spare cells, columns) with the product tree forced on and off: always the
    truth.  Shapes the fixed cases do not name: 2^6 coefficients (a tree of one leaf), blow-up 0 .. 5, cells larger than the polynomial."""
    rng = np.random.default_rng(424242)
    for case in range(60):
        L = int(rng.integers(1, 13))
        B = int(rng.integers(0 if L >= 2 else 1, 6))
        n = max(L + B, 2)
        m = int(rng.choice([0, 0, 0, 1, 2, 3, 5])) if n >= 3 else 0
        m = min(m, n - 1)
        ncols = int(rng.choice([1, 2, 3, 4, 5]))
        need = ((1 << max(L - m, 0)) + 1) if m > 0 else (1 << L) + 2
        total = 1 << (n - m)
        if need > total:
            continue  # (blow-up 0 leaves no spare samples)
        n_cells = int(min(total, need + rng.integers(0, 6)))
        coef = rng.integers(0, P, (ncols, 1 << L), dtype=np.uint32)
        ev = oracle.circle_evaluate(coef, n)
        idx = rng.permutation(total)[:n_cells].astype(np.uint32)
        cells = _cells(ev, idx, m)
        d_cells, d_c = DevBuf.from_array(gpu_ctx, cells), DevBuf(gpu_ctx, 4 * ncols << L)
        for min_log in ("6", "32"):
            gpu_ctx.set_option("FRIEDA_ERASURE_TREE_MIN_LOG", int(min_log))
            rc = gpu_ctx._L.frieda_circle_interpolate_points(gpu_ctx._h, d_cells.ptr, idx.ctypes.data, n_cells, ncols, m, L, n, d_c.ptr)
            assert rc == 0, (case, L, n, m, ncols, n_cells, min_log, gpu_ctx._L.frieda_last_error(gpu_ctx._h))
            assert np.array_equal(d_c.to_array(np.uint32, (ncols, 1 << L)), coef), (case, L, n, m, ncols, n_cells, min_log)
    gpu_ctx.set_option("FRIEDA_ERASURE_TREE_MIN_LOG", 15)


def _encode_on_device(gpu_ctx, data, B):
    L_ = gpu_ctx._L
    nf, npad, lg = C.c_size_t(), C.c_size_t(), C.c_uint32()
    L_.frieda_codec_shape(len(data), C.byref(nf), C.byref(npad), C.byref(lg))
    L, n = lg.value, lg.value + B
    d_in = DevBuf.from_array(gpu_ctx, np.frombuffer(data, dtype=np.uint8))
    d_coef, d_ev = DevBuf(gpu_ctx, 4 * npad.value), DevBuf(gpu_ctx, 16 << n)
    _check(gpu_ctx, L_.frieda_unpack30(gpu_ctx._h, d_in.ptr, len(data), d_coef.ptr, npad.value))
    _check(gpu_ctx, L_.frieda_circle_evaluate(gpu_ctx._h, d_coef.ptr, 4, L, n, d_ev.ptr))
    return d_ev.to_array(np.uint32, (4, 1 << n)), L, n


def test_reference_blob_from_single_sampled_points(gpu_ctx, blob):
    """The reference's own fixture (/root/reference/blob, 262 146 bytes -> 2^15 coefficients per column, 2^19-point codeword at the
    benches' blow-up): 2^15 + 2 single points sampled anywhere in the codeword give the blob back, byte for byte.  (Round 2 stopped at
    4096 cells, i.e. blobs of 61 KB for single points.)"""
    ev, L, n = _encode_on_device(gpu_ctx, blob, 4)
    assert (L, n) == (15, 19)
    rng = np.random.default_rng(2026)
    for n_pts in ((1 << L) + 2, (1 << L) + 3, (1 << L) + 5000, 1 << (n - 1)):  # the spare ones only serve the consistency check
        idx = rng.permutation(1 << n)[:n_pts].astype(np.uint32)
        cells = _cells(ev, idx, 0)
        assert gpu_ctx.reconstruct_from_points(cells, idx, L, n, len(blob)) == blob, n_pts


@pytest.mark.parametrize("n_bytes,B,m,extra_cells", [(3000, 2, 0, 2), (70001, 2, 1, 1), (983040, 4, 2, 9), (983040, 1, 0, 2), (3932160, 4, 0, 2),
                                                     (3932160, 4, 6, 33), (61440, 7, 0, 2), (15728640, 4, 0, 2),
                                                     (3932160, 4, 1, 3), (15728640, 4, 4, 5),  # many small cells: product-tree route
                                                     (62914560, 1, 0, 2)])  # 2^22 coefficients: 65536 leaves, the tree's transforms in two column chunks
def test_encode_sample_points_reconstruct_round_trip(gpu_ctx, n_bytes, B, m, extra_cells):
    """encode -> a sampling client's view (cells of 2^m entries scattered over the whole codeword, a handful more than the minimum) ->
    the original bytes; cell counts far beyond the 4096 of the dense solver (up to 2^20 + 2 single points on the 2^24 domain of the bench
    workload: 15.7 MB back from a 1/16 sample of its codeword)."""
    data = splitmix64_bytes(31 + m, n_bytes).tobytes()
    ev, L, n = _encode_on_device(gpu_ctx, data, B)
    m = min(m, L)
    rng = np.random.default_rng(n_bytes + B)
    n_cells = (1 << (L - m)) + extra_cells
    idx = rng.permutation(1 << (n - m))[:n_cells].astype(np.uint32)
    cells = _cells(ev, idx, m) if n_cells < 70000 else np.ascontiguousarray(ev.reshape(4, -1, 1 << m)[:, idx, :].transpose(1, 0, 2))
    assert gpu_ctx.reconstruct_from_points(cells, idx, L, n, n_bytes) == data


@pytest.mark.parametrize("spec,B,nq", [("pattern:1024", 4, 20), ("pattern:4096", 4, 20), ("blob", 4, 300), ("pattern:300", 2, 12)])
def test_das_loop_prove_verify_pool_reconstruct(gpu_ctx, blob, spec, B, nq):
    """The README's flow with what /root/reference/src actually has (README.md:56-69; src/proof.rs:32-101): a sample IS a proof — the
    seed of generate_proof decides through the transcript which positions are opened.  Sampling clients with different seeds each get a
    proof, verify it (frieda_verify_samples also says WHERE it sampled), and the pooled verified (position, value) pairs rebuild the
    blob once there are 2^L + 2 distinct ones.  Every returned position is checked against the codeword itself."""
    import frieda_amd
    from util import resolve_input

    data = resolve_input(spec, blob)
    ev, L, n = _encode_on_device(gpu_ctx, data, B)
    cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(B, 0, nq), 8)
    root = gpu_ctx.commit(data, B)
    pool = {}
    seed = 0
    while len(pool) < (1 << L) + 2:
        seed += 1
        assert seed < 20000, "the pool does not fill"
        commitment, proof = gpu_ctx.commit_and_generate_proof(data, seed, cfg)  # (the storage node's side)
        assert commitment == root
        ok, positions = frieda_amd.verify_samples(proof, seed)  # (the client's side)
        assert ok and frieda_amd.verify(proof, seed)
        evals = proof.evaluations
        assert len(positions) == len(evals) and np.all(np.diff(positions.astype(np.int64)) > 0) and positions.max() < (1 << n)
        assert np.array_equal(ev[:, positions].T, evals), "a returned position does not hold the proof's evaluation"
        for p, v in zip(positions.tolist(), evals):
            pool[p] = v
        if seed == 1:  # a wrong seed is a rejected proof and yields no positions
            bad_ok, bad_pos = frieda_amd.verify_samples(proof, seed + 1)
            assert not bad_ok and bad_pos is None
    idx = np.array(sorted(pool), dtype=np.uint32)
    cells = np.ascontiguousarray(np.stack([pool[int(p)] for p in idx]).astype(np.uint32).reshape(-1, 4, 1))
    assert gpu_ctx.reconstruct_from_points(cells, idx, L, n, len(data)) == bytes(data)
