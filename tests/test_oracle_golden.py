"""CPU: the oracle against every known answer the reference's tests hold for this path and the
reference's 13 test behaviours.  These pin the checker before it is trusted to check the HIP path."""
import hashlib

import numpy as np
import pytest

from conftest import pattern_bytes
from util import load_vectors, resolve_input

P = 2**31 - 1


def test_bytes_to_one_felt(oracle):
    # src/utils.rs:41-49
    for i in range(256):
        f = oracle.bytes_to_felt_le(bytes([i]))
        assert f.tolist() == [i]


def test_bytes_to_two_felt(oracle):
    # src/utils.rs:52-66: two 30-bit copies of i packed LSB-first into 8 bytes -> [i, i, 0]
    for i in range(513):
        bits = (i & (2**30 - 1)) | ((i & (2**30 - 1)) << 30)
        data = bits.to_bytes(8, "little")
        assert oracle.bytes_to_felt_le(data).tolist() == [i, i, 0]


def test_codec_against_python_bigint(oracle):
    rng = np.random.default_rng(7)
    for n in [0, 1, 3, 4, 15, 16, 29, 30, 31, 58, 119, 120, 121, 1000, 4097]:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        v = int.from_bytes(data, "little")
        exp = [(v >> (30 * k)) & (2**30 - 1) for k in range((8 * n + 29) // 30)]
        assert oracle.bytes_to_felt_le(data).tolist() == exp


def test_padding_rule(oracle):
    # src/utils.rs:23-27: next power of two >= 4, split in 4
    for n, (fp, L) in {0: (4, 0), 1: (4, 0), 15: (4, 0), 16: (8, 1), 58: (16, 2), 1024: (512, 7), 262146: (131072, 15)}.items():
        coef, lg = oracle.polynomial_from_bytes(bytes(n))
        assert coef.size == fp and lg == L


@pytest.mark.parametrize("vec", load_vectors()["commit"], ids=lambda v: v["input"][:24])
def test_commit_known_answers(oracle, blob, vec):
    # the first entry is the reference's golden root (src/commit.rs:28-38)
    data = resolve_input(vec["input"], blob)
    assert oracle.commit(data, vec["log_blowup_factor"]).hex() == vec["root"]


def test_evaluation_heads(oracle):
    heads = load_vectors()["evaluation_heads"]
    coef, L = oracle.polynomial_from_bytes(pattern_bytes(1024).tobytes())
    ev = oracle.circle_evaluate(coef, L + 4)
    assert ev[0, :4].tolist() == heads["pattern:1024"]["col0"]
    s = b"This is the original data that needs to be made available."
    coef, L = oracle.polynomial_from_bytes(s)
    ev = oracle.circle_evaluate(coef, L + 4)
    assert ev[:, :2].tolist() == heads["ascii:" + s.decode()]["cols"]


def test_fft_matches_direct_basis_evaluation(oracle):
    """Independent check of A.2: out[i] = f(domain.at(brev(i))) with the circle-FFT basis y^j0 x^j1 pi(x)^j2 ..."""
    import ctypes as C

    rng = np.random.default_rng(3)
    for n in (3, 4, 5):
        coef = rng.integers(0, P, (1, 1 << n), dtype=np.uint32)
        ev = oracle.circle_evaluate(coef, n)[0]
        x, y = C.c_uint32(), C.c_uint32()
        for i in range(1 << n):
            oracle.lib().fo_circle_domain_at(n, oracle.lib().fo_bit_reverse_index(i, n), C.byref(x), C.byref(y))
            px, py = x.value, y.value
            acc = 0
            for j in range(1 << n):
                term = int(coef[0, j])
                if j & 1:
                    term = term * py % P
                xx = px
                for b in range(1, n):
                    if (j >> b) & 1:
                        term = term * xx % P
                    xx = (2 * xx * xx - 1) % P
                acc = (acc + term) % P
            assert acc == ev[i]


def test_standard_blake2s_matches_hashlib(oracle):
    import ctypes as C

    for n in [0, 1, 31, 32, 63, 64, 65, 127, 128, 129, 300]:
        d = bytes((7 * i + 3) % 256 for i in range(n))
        out = (C.c_uint8 * 32)()
        a = np.frombuffer(d, dtype=np.uint8)
        oracle.lib().fo_blake2s256(a.ctypes.data if n else None, n, out)
        assert bytes(out) == hashlib.blake2s(d).digest()


def test_merkle_hash_is_not_rfc_blake2s(oracle):
    """Blake2sMerkleHasher is the bare compression function, not Blake2s-256."""
    cols = np.arange(8, dtype=np.uint32).reshape(4, 2)
    leaves = oracle.merkle_commit_layer(1, None, cols)
    msg = np.array([0, 2, 4, 6] + [0] * 12, dtype="<u4").tobytes()
    assert bytes(leaves[0]) != hashlib.blake2s(msg[:16]).digest()
    assert bytes(leaves[0]) != hashlib.blake2s(msg).digest()


# ---- the reference's proof tests (src/proof.rs:119-193, src/lib.rs:52-85) on the oracle ----
PCS = dict(pow_bits=20, log_blowup_factor=4, log_last_layer_degree_bound=1, n_queries=20)


@pytest.fixture(scope="module")
def blob_proof(oracle, blob):
    return oracle.commit_and_generate_proof(blob, None, oracle.make_config(**PCS))


def test_generate_proof(blob_proof):
    assert blob_proof[1].c.n_inner_layers != 0  # src/proof.rs:119-124


def test_commit_and_generate_proof(oracle, blob, blob_proof):
    root, proof = blob_proof  # src/proof.rs:126-135
    assert root == oracle.commit(blob, 4)
    assert bytes(proof.c.first_layer.commitment) == root


def test_verify_proof(oracle, blob_proof):
    assert oracle.verify(blob_proof[1], None)


def test_verify_invalid_pow(oracle, blob_proof):
    p = blob_proof[1].clone()
    p.c.proof_of_work += 1
    assert not oracle.verify(p, None)


def _set_evals(p, ev):
    flat = np.ascontiguousarray(ev, dtype=np.uint32).ravel()
    for i, v in enumerate(flat):
        p.c.evaluations[i] = int(v)


def test_verify_invalid_evaluations(oracle, blob_proof):
    p = blob_proof[1].clone()
    ev = p.evaluations()
    ev[0] = (ev[0].astype(np.uint64) + 1) % P
    _set_evals(p, ev)
    assert not oracle.verify(p, None)


def test_verify_invalid_evaluations_order(oracle, blob_proof):
    p = blob_proof[1].clone()
    _set_evals(p, p.evaluations()[::-1])
    assert not oracle.verify(p, None)


def test_verify_invalid_evaluations_length_panics(oracle, blob_proof):
    p = blob_proof[1].clone()  # src/proof.rs:166-173 #[should_panic]
    p.c.n_evaluations -= 1
    with pytest.raises(RuntimeError):
        oracle.verify(p, None)
    p.c.n_evaluations += 1


def test_verify_invalid_1_evaluation_unordered(oracle, blob_proof):
    p = blob_proof[1].clone()
    ev = p.evaluations()
    ev[[0, 1]] = ev[[1, 0]]
    _set_evals(p, ev)
    assert not oracle.verify(p, None)


def test_verify_with_seed(oracle, blob):
    cfg = oracle.make_config(**PCS)
    _, p1 = oracle.commit_and_generate_proof(blob, 1, cfg)
    _, p2 = oracle.commit_and_generate_proof(blob, 2, cfg)
    assert p1.evaluations().tolist() != p2.evaluations().tolist()
    assert oracle.verify(p1, 1) and oracle.verify(p2, 2)
    assert not oracle.verify(p1, 2) and not oracle.verify(p2, 1)


def test_end_to_end(oracle):
    s = b"This is the original data that needs to be made available."  # src/lib.rs:52-85
    cfg = oracle.make_config(20, 4, 0, 20)
    root, proof = oracle.commit_and_generate_proof(s, None, cfg)
    assert root == oracle.commit(s, 4)
    assert oracle.verify(proof, None)


@pytest.mark.parametrize("n_bytes,B,last", [(1024, 4, 0), (1024, 4, 1), (4096, 4, 0), (300, 2, 1), (2000, 1, 2), (700, 3, 0)])
def test_fri_degree_invariant(oracle, n_bytes, B, last):
    """stwo asserts the last layer interpolates below the degree bound; the prover returns the panic status if not."""
    data = pattern_bytes(n_bytes).tobytes()
    cfg = oracle.make_config(8, B, last, 12)
    _, proof = oracle.commit_and_generate_proof(data, 5, cfg)
    assert proof.c.n_last_layer_poly == 1 << last
    assert oracle.verify(proof, 5)


def test_too_small_polynomial_panics(oracle):
    with pytest.raises(RuntimeError):
        oracle.commit_and_generate_proof(b"tiny", None, oracle.make_config(8, 4, 0, 4))


@pytest.mark.parametrize("L,n,m", [(1, 2, 1), (3, 5, 1), (3, 5, 3), (4, 8, 2), (6, 10, 3), (8, 12, 4), (8, 9, 2), (10, 14, 5)])
def test_oracle_reconstruct_from_scattered_cells(oracle, L, n, m):
    """evaluate -> keep R = 2^(L-m) random cells of 2^m consecutive (bit-reversed) entries -> reconstruct == the coefficients.
    (m == L is the single-block case of fo_circle_interpolate_block.)"""
    rng = np.random.default_rng(1000 * L + 10 * n + m)
    P = (1 << 31) - 1
    coef = rng.integers(0, P, size=(4, 1 << L), dtype=np.uint32)
    ev = oracle.circle_evaluate(coef, n)
    R = 1 << (L - m)
    for trial in range(3):
        idx = rng.choice(1 << (n - m), size=R, replace=False).astype(np.uint32)
        cells = np.stack([ev[:, int(c) << m : (int(c) + 1) << m] for c in idx])  # [R, 4, 2^m]
        got = oracle.reconstruct_cells(cells, idx, n, L)
        assert np.array_equal(got, coef), (trial, idx[:4])
    with pytest.raises(ValueError):
        oracle.reconstruct_cells(cells, np.zeros(R, dtype=np.uint32) if R > 1 else np.array([1 << (n - m)], dtype=np.uint32), n, L)


@pytest.mark.parametrize("L,n,m", [(2, 4, 1), (3, 5, 2), (3, 5, 1)])
def test_oracle_every_set_of_distinct_cells_reconstructs(oracle, L, n, m):
    """Exhaustive over ALL subsets of 2^(L-m) cells for small shapes: none is singular, each gives back the coefficients."""
    import itertools

    rng = np.random.default_rng(7)
    coef = rng.integers(0, (1 << 31) - 1, size=(1, 1 << L), dtype=np.uint32)
    ev = oracle.circle_evaluate(coef, n)
    tw, itw = oracle.precompute_twiddles(n)
    for cs in itertools.combinations(range(1 << (n - m)), 1 << (L - m)):
        cells = np.stack([ev[:, c << m : (c + 1) << m] for c in cs])
        assert np.array_equal(oracle.reconstruct_cells(cells, np.array(cs, dtype=np.uint32), n, L, tw, itw), coef), cs


def test_bit_reverse_column_is_the_index_permutation(oracle):
    """ColumnOps::bit_reverse_column (stwo core/utils.rs bit_reverse): out[brev(i)] = in[i]."""
    for lg in (0, 1, 2, 5, 9, 12):
        v = np.arange(1 << lg, dtype=np.uint32) * 7 + 3
        got = oracle.bit_reverse_column(v)
        idx = np.array([int(format(i, f"0{lg}b")[::-1], 2) if lg else 0 for i in range(1 << lg)])
        assert np.array_equal(got[idx], v)
        assert np.array_equal(oracle.bit_reverse_column(got), v)


def test_selfcheck_trace_is_reproduced_by_the_oracle(oracle):
    """tests/golden/trace_selfcheck.json is SELF-GENERATED (tools/dump_trace.py), not reference-held: this only guards the oracle
    against drifting away from the file a cargo-side diff would be made against."""
    import json
    import os
    import sys

    from conftest import GOLDEN, ROOT

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dump_trace

    doc = json.load(open(os.path.join(GOLDEN, "trace_selfcheck.json")))
    assert "NOT reference-held" in doc["provenance"]
    fresh = dump_trace.build_doc()
    assert fresh["cases"] == doc["cases"]
    # the tie the reference does test (src/proof.rs:126-135): first FRI root == commit() root, here the golden root for the blob
    blob_case = [c for c in doc["cases"] if c["input"] == "blob"][0]
    assert blob_case["roots"][0] == blob_case["commitment"] == "d1a2d5069dc587e55dc29cc6255af937ff7fed0ee41bdf5af98717f9d74f60e8"


@pytest.mark.parametrize("L,n,m", [(0, 1, 0), (1, 1, 0), (2, 2, 0), (3, 5, 0), (5, 9, 0), (4, 6, 2), (6, 8, 1)])
def test_oracle_reconstructs_from_scattered_cells_and_points(oracle, L, n, m):
    """fo_reconstruct_cells: any 2^(L-m) distinct cells of 2^m entries give the coefficients back; m == 0 = single sampled points
    (the circle layer's twiddle enters the system).  Pinned through the transform itself (fo_circle_evaluate is golden-root pinned)."""
    rng = np.random.default_rng(50 + 10 * L + n + m)
    coef = rng.integers(0, 2**31 - 1, (2, 1 << L), dtype=np.uint32)
    ev = oracle.circle_evaluate(coef, n)
    R = 1 << (L - m)
    solved = 0
    for _ in range(12):
        idx = rng.choice(1 << (n - m), size=R, replace=False).astype(np.uint32)
        cells = np.ascontiguousarray(np.stack([ev[:, int(c) << m : (int(c) + 1) << m] for c in idx]))
        try:
            got = oracle.reconstruct_cells(cells, idx, n, L)
        except ValueError:
            # Cells of >= 2 entries hold whole conjugate pairs, the system splits into two Vandermonde systems in x and is never
            # singular.  Single points can be: the 2^L-dimensional code space is one dimension short of the Riemann-Roch space,
            # so a point set fails when the unique function vanishing on it falls into the code space (a few % of the sets at
            # 2^5-point domains, ~1/P at real sizes).  The solver must say so — never return a wrong polynomial.
            assert m == 0
            continue
        assert np.array_equal(got, coef)
        solved += 1
    assert solved >= 6


def test_known_singular_point_set_is_reported(oracle):
    """8 of the 32 points of a 2^5 domain that do NOT determine a 2^3-coefficient polynomial (found by search): the solver says so."""
    coef = np.arange(1, 9, dtype=np.uint32).reshape(1, 8) * 1234567
    ev = oracle.circle_evaluate(coef, 5)
    sing = np.array([1, 3, 4, 8, 11, 14, 16, 27], dtype=np.uint32)
    with pytest.raises(ValueError):
        oracle.reconstruct_cells(np.ascontiguousarray(ev[:, sing].T.reshape(-1, 1, 1)), sing, 5, 3)
    ok = np.array([1, 3, 4, 8, 11, 14, 16, 30], dtype=np.uint32)  # one point exchanged
    assert np.array_equal(oracle.reconstruct_cells(np.ascontiguousarray(ev[:, ok].T.reshape(-1, 1, 1)), ok, 5, 3), coef)


@pytest.mark.parametrize("L,n,extra", [(1, 3, 2), (2, 4, 2), (3, 6, 3), (4, 8, 5), (5, 7, 2), (6, 10, 40), (7, 8, 120)])
def test_reconstruct_points_matches_dense_solve_and_round_trip(oracle, L, n, extra):
    """fo_reconstruct_points (any >= 2^L + 2 sampled points; erasure-locator route, the restatement frieda_circle_interpolate_points is
    compared with on the GPU) against the truth and against the independent dense solve of fo_reconstruct_cells on the same samples."""
    rng = np.random.default_rng(9000 + 10 * L + n)
    P = 2**31 - 1
    coef = rng.integers(0, P, (2, 1 << L), dtype=np.uint32)
    ev = oracle.circle_evaluate(coef, n)
    pos = rng.permutation(1 << n)[: (1 << L) + extra].astype(np.uint32)
    vals = np.ascontiguousarray(ev[:, pos].T)
    assert np.array_equal(oracle.reconstruct_points(vals, pos, n, L), coef)
    # repeated positions are ignored; one point short is refused
    assert np.array_equal(oracle.reconstruct_points(np.concatenate([vals, vals[:2]]), np.concatenate([pos, pos[:2]]), n, L), coef)
    with pytest.raises(ValueError):
        oracle.reconstruct_points(vals[: (1 << L) + 1], pos[: (1 << L) + 1], n, L)
    bad = vals.copy()
    bad[-1, 0] ^= 1  # a corrupted sample (the last one: a spare whenever extra > 2) is reported, not absorbed
    with pytest.raises(ValueError, match="not values of one polynomial"):
        oracle.reconstruct_points(bad, pos, n, L)
    if L >= 2:
        try:
            dense = oracle.reconstruct_cells(np.ascontiguousarray(ev[:, pos[: 1 << L]].T.reshape(1 << L, 2, 1)), pos[: 1 << L], n, L)
        except ValueError:
            return  # (a singular set of exactly 2^L single points: the dense route's own limitation)
        assert np.array_equal(dense, coef)
