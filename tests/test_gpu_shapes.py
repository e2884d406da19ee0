"""GPU: the encode's pass planner (frieda_amd/csrc/ntt.hip, circle_evaluate_into_tree) at MB-scale blob sizes — every branch, against the
oracle, bit-exact.

`polynomial_from_bytes` (/root/reference/src/utils.rs:21-33) maps ANY byte length to a power-of-two polynomial: L = log2 of the
coefficients per column, n = L + log_blowup_factor (/root/reference/src/commit.rs:14-16).  The planner splits the L real layers into a
contiguous last pass of min(L, 12) layers and strided passes over the `rest = L - 12` layers above it, and picks a kernel per pass:

  * `last12`              rest == 0 (L <= 12): only the contiguous pass (reads the coefficients themselves, replicated)
  * `fast(4)`, `fast(8)`, `fast(4+8)`, `fast(8+8)` ...
                          the strided layers as passes of the fast kernel — 8 layers (two radix-16 stages) or 4 layers (one stage), a
                          4-layer pass first when the number of 4-layer units is odd — with `pad=k`: rest rounded up to a multiple of 4 by
                          k = 1..3 zero-padded layers, which the first pass executes against zero coefficients (needs k <= log_blowup_factor)
  * `generic(t,w)`        k > log_blowup_factor, rest < 8: one strided pass of t layers through the generic kernel with runs of 2^w words
  * `two(t1+t2)`          k > log_blowup_factor, rest > 8: two generic strided passes

`encode_plan` below mirrors that decision so that every test id names the branch it reaches; the product is not told which branch to
take (the mirror is only used for the ids and to assert that the matrix covers all of them).
"""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import splitmix64_bytes
from util import DevBuf

pytestmark = pytest.mark.gpu

P = 2**31 - 1


def encode_plan(L, B):
    """Name of the planner branch for 2^L coefficients per column at blow-up 2^B (mirror of ntt.hip's decision, ids only)."""
    if L <= 12:
        return "last12"
    rest = L - 12
    units = (rest + 3) // 4
    pad = 4 * units - rest
    if pad <= B:
        passes = (["4"] if units & 1 else []) + ["8"] * (units // 2)
        return f"fast({'+'.join(passes)}{',pad=%d' % pad if pad else ''})"
    if rest < 8:
        return f"generic(t={rest},w={12 - rest})"
    t1 = (rest + 1) // 2
    return f"two({t1}+{rest - t1})"


def exact_len(L):
    """Bytes whose 30-bit felts exactly fill 4 columns of 2^L coefficients."""
    return (4 << L) * 30 // 8


def shape_lengths(L):
    """exact fill; a ragged length inside the same L; the smallest length with that L (half + 1 byte: columns 2 and 3 all zero)."""
    e = exact_len(L)
    return {"exact": e, "ragged": e - 12345, "halfplus1": e // 2 + 1}


COMMIT_CASES = []
for _L in (16, 17, 18, 19, 20, 21):
    for _B in (1, 2, 4):
        for _kind, _len in shape_lengths(_L).items():
            if _L == 21 and _B == 4 and _kind == "halfplus1":
                continue  # 2^25 domain: two lengths are enough (the oracle needs ~25 s each)
            COMMIT_CASES.append(pytest.param(_L, _B, _len, id=f"L{_L}-B{_B}-{encode_plan(_L, _B)}-{_kind}"))
# rest 1..3 (a single padded 4-layer pass, or the generic kernel when the blow-up is smaller than the padding), large blow-ups, L = 22
for _L, _B in ((13, 7), (14, 6), (15, 5), (13, 2), (14, 1), (15, 1), (22, 2), (22, 1)):
    COMMIT_CASES.append(pytest.param(_L, _B, exact_len(_L) - 777, id=f"L{_L}-B{_B}-{encode_plan(_L, _B)}-ragged"))


def test_matrix_reaches_every_planner_branch():
    seen = {c.id.split("-")[2] for c in COMMIT_CASES} | {f"{encode_plan(L, n - L)}" for L, n in EVAL_CASES}
    for want in ("fast(4)", "fast(8)", "fast(4+8)", "fast(8,pad=1)", "fast(8,pad=2)", "fast(8,pad=3)", "fast(4,pad=1)", "fast(4,pad=2)", "fast(4,pad=3)",
                 "fast(4+8,pad=3)", "fast(4+8,pad=2)", "generic(t=5,w=7)", "generic(t=6,w=6)", "two(5+4)"):
        assert want in seen, (want, sorted(seen))
    for L in (16, 17, 18, 19, 20, 21):
        fp = 4
        F = (8 * shape_lengths(L)["halfplus1"] + 29) // 30
        while fp < F:
            fp *= 2
        assert fp == 4 << L  # half + 1 byte still has 2^L coefficients per column


@pytest.fixture(scope="module")
def oracle_roots(oracle):
    """commit() roots of every case from the CPU oracle, computed on a few host threads (the C oracle releases the GIL)."""
    keys = sorted({(c.values[2], c.values[1], c.values[0]) for c in COMMIT_CASES}, key=lambda k: -(k[2] + k[1]))

    def one(k):
        length, B, L = k
        return k, oracle.commit(splitmix64_bytes(1000 + L * 8 + B, length).tobytes(), B)

    with ThreadPoolExecutor(max_workers=6) as ex:  # <= 6 x 2.7 GB of oracle workspace at the 2^25 domain
        return dict(ex.map(one, keys))


@pytest.mark.parametrize("L,B,length", COMMIT_CASES)
def test_commit_root_mb_scale_shapes(gpu_ctx, oracle, oracle_roots, L, B, length):
    """commit() (/root/reference/src/commit.rs:11-22) at blob sizes 0.5 .. 8 MB, blow-ups 2, 4, 16: root equal to the oracle's."""
    data = splitmix64_bytes(1000 + L * 8 + B, length)
    coef, lg = oracle.polynomial_from_bytes(data[: min(length, 64)])  # cheap sanity of the helper itself
    nf, npad, lgs = C.c_size_t(), C.c_size_t(), C.c_uint32()
    gpu_ctx._L.frieda_codec_shape(length, C.byref(nf), C.byref(npad), C.byref(lgs))
    assert lgs.value == L, "the length does not have the shape its id claims"
    assert gpu_ctx.commit(data.tobytes(), B) == oracle_roots[(length, B, L)]


PROVE_CASES = [
    # (L, B, length kind, pow, last, queries)
    (17, 4, "exact", 12, 0, 20),  # 2^21 domain
    (19, 4, "ragged", 10, 0, 20),  # 2^23 domain
    (17, 2, "ragged", 8, 1, 33),  # generic single pass (pad 3 > B)
    (18, 1, "exact", 8, 0, 20),  # generic single pass (pad 2 > B), 2^19 domain
    (21, 1, "ragged", 6, 2, 20),  # two strided passes, 2^22 domain
    (16, 3, "halfplus1", 8, 0, 20),  # generic (pad 4 > B), 2^19 domain
    (14, 6, "ragged", 6, 0, 20),  # one 4-layer pass, pad 2
    (15, 5, "exact", 6, 3, 20),  # one 4-layer pass, pad 1
    (13, 7, "ragged", 6, 0, 20),  # one 4-layer pass, pad 3
    (13, 2, "ragged", 6, 0, 20),  # generic (pad 3 > B)
    (21, 3, "exact", 6, 0, 20),  # 4-layer pass (pad 3: one real layer) + 8-layer pass, 2^24 domain
]


@pytest.mark.parametrize("L,B,kind,pow_bits,last,nq", PROVE_CASES, ids=[f"L{c[0]}-B{c[1]}-{encode_plan(c[0], c[1])}-{c[2]}" for c in PROVE_CASES])
def test_whole_proof_mb_scale_shapes(gpu_ctx, oracle, L, B, kind, pow_bits, last, nq):
    """commit_and_generate_proof (/root/reference/src/proof.rs:32-77) on odd domains and ragged MB-scale lengths: the whole proof
    byte-identical to the oracle's (the first tree after the fused kernel starts from 2^(n-6) hashes with n odd as well as even)."""
    import frieda_amd

    length = shape_lengths(L)[kind]
    data = splitmix64_bytes(2000 + L * 8 + B, length).tobytes()
    cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(B, last, nq), pow_bits)
    g_root, g_proof = gpu_ctx.commit_and_generate_proof(data, length, cfg)
    o_root, o_proof = oracle.commit_and_generate_proof(data, length, oracle.make_config(pow_bits, B, last, nq))
    assert g_root == o_root
    assert g_proof.n_inner_layers == L + B - 1 - B - last
    assert g_proof.serialize() == o_proof.serialize()
    assert frieda_amd.verify(g_proof, length)


EVAL_CASES = [(17, 21), (19, 23), (17, 18), (19, 20), (21, 25), (21, 22), (18, 19), (16, 17), (22, 23), (20, 24), (22, 24), (24, 25), (16, 20), (13, 14), (14, 16)]


@pytest.mark.parametrize("ncols", [1, 4])
@pytest.mark.parametrize("L,n", EVAL_CASES, ids=[f"L{L}-n{n}-{encode_plan(L, n - L)}" for L, n in EVAL_CASES])
def test_circle_evaluate_mb_scale_shapes(gpu_ctx, oracle, L, n, ncols):
    """Level B `PolyOps::evaluate` (frieda_circle_evaluate) per column on the same shapes, 1 and 4 columns, every output word."""
    if n == 25 and ncols == 4:
        ncols = 2  # two columns share a workgroup (the 4-column shape is covered by commit() at L = 21, B = 4)
    rng = np.random.default_rng(500 + 32 * L + n)
    coef = rng.integers(0, P, (ncols, 1 << L), dtype=np.uint32)
    tw, _ = oracle.precompute_twiddles(n)
    with ThreadPoolExecutor(max_workers=4) as ex:
        exp = np.concatenate(list(ex.map(lambda c: oracle.circle_evaluate(coef[c : c + 1], n, tw), range(ncols))))
    d_c = DevBuf.from_array(gpu_ctx, coef)
    d_o = DevBuf(gpu_ctx, 4 * ncols << n)
    from frieda_amd.api import _check

    _check(gpu_ctx._L.frieda_circle_evaluate(gpu_ctx._h, d_c.ptr, ncols, L, n, d_o.ptr), gpu_ctx._h)
    got = d_o.to_array(np.uint32, (ncols, 1 << n))
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("knob", ["FRIEDA_NO_ENCODE_TREE_FUSION", "FRIEDA_ENCODE_TREE_FUSION_PROVE", "FRIEDA_NTT_NO_PAD8", "FRIEDA_NTT_TREE_REG_ONLY", "FRIEDA_T5_REG3_LOG=18", "FRIEDA_NTT_CPW=2",
                                  "FRIEDA_HOST_DECOMMIT", "FRIEDA_NO_SMALL_FUSED", "FRIEDA_NTT_CPW_SMALL=4", "FRIEDA_NTT_REP", "FRIEDA_NTT_REP=0", "FRIEDA_TAIL_RUN_LOG=6",
                                  "FRIEDA_T9_MAX_LOG=12", "FRIEDA_UNPACK_TILES=1", "FRIEDA_T5_WIDE_LOG=16", "FRIEDA_TP_MIN_WGS=0", "FRIEDA_TP_MIN_WGS=1073741824", "FRIEDA_TREE_SKIP_LOG=10", "FRIEDA_TREE_SKIP_LONE_LOG=10", "FRIEDA_TREE_SKIP_LOG=40"])
def test_knob_variants_on_their_own_context(oracle, knob):
    """The A/B options of DESIGN.md §10 select other kernels / templates for the same result (unfused encode + leaf launch, generic
    strided pass instead of the padded 8-layer one, the register-only tree variants, the compressions' throughput form — runs with
    switched wave priority — in EVERY tree launch or in none, ...).  They are PER CONTEXT
    (frieda_ctx_set_option; the environment only sets a new context's defaults), so each variant runs on its own context in this
    process: commit root and whole proof against the oracle on five shapes."""
    import frieda_amd

    name, _, val = knob.partition("=")
    ctx = frieda_amd.Context(0)
    try:
        ctx.set_option(name, int(val or "1"))
        for L, B in ((16, 4), (18, 2), (13, 7), (12, 4), (7, 4)):
            length = (4 << L) * 30 // 8 - 4321 if L > 8 else (4 << L) * 30 // 8 - 7
            data = splitmix64_bytes(3000 + L, length).tobytes()
            assert ctx.commit(data, B) == oracle.commit(data, B), ("commit", L, B)
            cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(B, 0, 40), 6)
            r, p = ctx.commit_and_generate_proof(data, 9, cfg)
            o_r, o_p = oracle.commit_and_generate_proof(data, 9, oracle.make_config(6, B, 0, 40))
            assert r == o_r and p.serialize() == o_p.serialize(), ("prove", L, B)
    finally:
        ctx.close()


@pytest.mark.parametrize("host_decommit", [0, 1])
def test_unstored_tree_levels_are_rehashed_or_rebuilt(oracle, host_decommit):
    """A proof's large trees do not keep the two levels above their leaves (FRIEDA_TREE_SKIP_LOG, default 2^18 leaves): the device
    decommitment re-hashes the opened nodes of those levels from the layer's values, and the host planner (FRIEDA_HOST_DECOMMIT, or the
    device kernel's overflow exit) rebuilds such trees before it gathers.  Threshold lowered to 2^10 so that every layer of these shapes
    above the single-workgroup tail takes the new path; 20 and 300 queries (duplicates, many opened paths); whole proofs against the oracle."""
    import frieda_amd

    ctx = frieda_amd.Context(0)
    try:
        ctx.set_option("FRIEDA_TREE_SKIP_LOG", 10)
        ctx.set_option("FRIEDA_TREE_SKIP_LONE_LOG", 10)  # (a call of one blob has its own, higher threshold)
        ctx.set_option("FRIEDA_HOST_DECOMMIT", host_decommit)
        for L, B, nq in ((16, 4, 20), (14, 3, 300), (12, 4, 64), (17, 1, 20)):
            length = (4 << L) * 30 // 8 - 1234
            data = splitmix64_bytes(4100 + L, length).tobytes()
            cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(B, 0, nq), 5)
            r, p = ctx.commit_and_generate_proof(data, 11, cfg)
            o_r, o_p = oracle.commit_and_generate_proof(data, 11, oracle.make_config(5, B, 0, nq))
            assert r == o_r and p.serialize() == o_p.serialize(), (L, B, nq, host_decommit)
        # a batch: every blob's own workspace offset in the re-hash and in the rebuild
        blobs = [splitmix64_bytes(4200 + i, (4 << 14) * 30 // 8).tobytes() for i in range(3)]
        cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 5)
        got = ctx.commit_and_generate_proof_batch(blobs, [7, 8, 9], cfg)
        for (r, p), blob, seed in zip(got, blobs, [7, 8, 9]):
            o_r, o_p = oracle.commit_and_generate_proof(blob, seed, oracle.make_config(5, 4, 0, 20))
            assert r == o_r and p.serialize() == o_p.serialize()
    finally:
        ctx.close()


def test_options_are_per_context_and_checked(gpu_ctx):
    import frieda_amd

    with pytest.raises(frieda_amd.FriedaError):
        gpu_ctx.set_option("FRIEDA_NO_SUCH_OPTION", 1)
    with pytest.raises(frieda_amd.FriedaError):
        gpu_ctx.set_option("FRIEDA_TAIL_RUN_LOG", 99)
    other = frieda_amd.Context(0)
    other.set_option("FRIEDA_TAIL_RUN_LOG", 5)  # leaves gpu_ctx (and every other context) alone: nothing to observe but no error
    other.close()
