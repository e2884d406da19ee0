"""CPU: the C-ABI library loads without a GPU and exports every symbol the header declares; the product's host-side logic
(codec shape, wire format, verifier) against the oracle.  No compute entry point is called here."""
import ctypes as C

import numpy as np
import pytest

from conftest import pattern_bytes

P = 2**31 - 1


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as g

    g.build()
    from frieda_amd import _lib

    return _lib.lib()


def test_library_exports_every_declared_symbol(L):
    from frieda_amd import _lib

    syms = _lib.declared_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(L, s), f"libfrieda_hip.so does not export {s}"
    # the ctypes signature table covers the header exactly
    assert sorted(L._signatures) == syms
    assert L.frieda_abi_version() == 1
    assert b"panic" in L.frieda_status_string(3)


def test_header_has_no_torch_or_cxx_types():
    from frieda_amd import _lib

    import re

    text = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER_PATH).read(), flags=re.S)  # declarations only, comments stripped
    for bad in ("torch", "at::", "std::", "template", "class ", "&"):
        assert bad not in text, bad


def test_product_never_imports_oracle():
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, files in os.walk(os.path.join(root, "frieda_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle|#include\s+\".*oracle", src, re.M), f


def test_codec_shape_matches_reference_f64_rule(L, oracle):
    nf, npad, lg = C.c_size_t(), C.c_size_t(), C.c_uint32()
    sizes = list(range(0, 600)) + [1023, 1024, 1025, 3839, 3840, 3841, 65536, 262146, 983040, 3932160, 15728640, 15728641]
    for n in sizes:
        L.frieda_codec_shape(n, C.byref(nf), C.byref(npad), C.byref(lg))
        f = oracle.lib().fo_felt_count(n)
        fp = oracle.lib().fo_padded_len(f)
        assert (nf.value, npad.value) == (f, fp), n
        assert (1 << (lg.value + 2)) == fp


def test_merkle_layer_offsets(L, oracle):
    for m in range(0, 20):
        for l in range(0, m + 1):
            assert L.frieda_merkle_layer_offset(m, l) == oracle.lib().fo_merkle_layer_offset(m, l)


def test_null_arguments_are_rejected(L):
    assert L.frieda_ctx_create(0, None, None) == 1
    assert L.frieda_commit(None, None, 0, 4, None) == 1
    assert L.frieda_verify(None, None, None) == 1
    assert L.frieda_proof_deserialize(None, 0, None) == 1


# ---- wire format + verifier against oracle-made proofs ----
@pytest.fixture(scope="module")
def proofs(oracle, blob):
    out = {}
    out["blob"] = oracle.commit_and_generate_proof(blob, None, oracle.make_config(20, 4, 1, 20))
    out["p1024"] = oracle.commit_and_generate_proof(pattern_bytes(1024).tobytes(), 1024, oracle.make_config(20, 4, 0, 20))
    out["small"] = oracle.commit_and_generate_proof(pattern_bytes(300).tobytes(), 9, oracle.make_config(8, 2, 1, 12))
    return out


def test_wire_image_round_trip_and_cross_verify(L, proofs):
    import frieda_amd

    for name, (root, op) in proofs.items():
        img = op.serialize()
        p = frieda_amd.Proof.deserialize(img)
        assert p.serialize() == img, name
        assert p.commitment == root
        assert p.proof_of_work == op.c.proof_of_work
        assert p.log_size_bound == op.c.log_size_bound
        assert np.array_equal(p.evaluations, op.evaluations())
        assert p.n_inner_layers == op.c.n_inner_layers
    seeds = {"blob": None, "p1024": 1024, "small": 9}
    for name, (root, op) in proofs.items():
        p = frieda_amd.Proof.deserialize(op.serialize())
        assert frieda_amd.verify(p, seeds[name])
        assert not frieda_amd.verify(p, 12345)


def test_verify_samples_returns_the_sampled_positions(oracle, proofs, blob):
    """frieda_verify_samples (host verifier): for an accepted proof, out_positions[i] is where evaluations[i] sits in the bit-reversed
    codeword — checked against the oracle's own encode of the same blob; a rejected proof yields none."""
    import frieda_amd

    inputs = {"blob": (blob, None, 4), "p1024": (pattern_bytes(1024).tobytes(), 1024, 4), "small": (pattern_bytes(300).tobytes(), 9, 2)}
    for name, (root, op) in proofs.items():
        data, seed, B = inputs[name]
        p = frieda_amd.Proof.deserialize(op.serialize())
        ok, pos = frieda_amd.verify_samples(p, seed)
        assert ok and len(pos) == len(p.evaluations) and np.all(np.diff(pos.astype(np.int64)) > 0)
        coef, lg = oracle.polynomial_from_bytes(data)
        ev = oracle.circle_evaluate(coef, lg + B)
        assert np.array_equal(ev[:, pos].T, p.evaluations), name
        assert frieda_amd.verify_samples(p, 777) == (False, None)
        # a buffer smaller than the number of positions is refused (and says how many there are); null arguments are refused
        import ctypes as C

        L_ = frieda_amd._lib.lib()
        ok, n_pos = C.c_int(0), C.c_size_t(0)
        small = np.zeros(2, dtype=np.uint32)
        sp = C.byref(C.c_uint64(seed)) if seed is not None else None
        assert L_.frieda_verify_samples(p._h, sp, C.byref(ok), small.ctypes.data, 2, C.byref(n_pos)) == 1 and n_pos.value == len(pos) and ok.value == 1
        assert L_.frieda_verify_samples(p._h, sp, None, small.ctypes.data, 2, C.byref(n_pos)) == 1
        assert L_.frieda_verify_samples(None, sp, C.byref(ok), small.ctypes.data, 2, C.byref(n_pos)) == 1


def test_malformed_images_are_rejected(L, proofs):
    import frieda_amd

    img = proofs["small"][1].serialize()
    for bad in (b"", img[:7], img[:-1], img + b"\x00", b"XXXX" + img[4:], img[:36] + b"\xff\xff\xff\xff" + img[40:]):
        with pytest.raises(frieda_amd.FriedaError):
            frieda_amd.Proof.deserialize(bad)


def test_product_verifier_matches_reference_rejections(proofs):
    """src/proof.rs:143-181 on the product's host verifier, fed with an oracle-made proof."""
    import frieda_amd

    base = frieda_amd.Proof.deserialize(proofs["blob"][1].serialize())
    assert frieda_amd.verify(base, None)
    p = base.clone()
    p.proof_of_work += 1
    assert not frieda_amd.verify(p, None)
    p = base.clone()
    e = p.evaluations
    e[0] = (e[0].astype(np.uint64) + 1) % P
    p.evaluations = e
    assert not frieda_amd.verify(p, None)
    p = base.clone()
    p.evaluations = p.evaluations[::-1]
    assert not frieda_amd.verify(p, None)
    p = base.clone()
    p.evaluations = p.evaluations[:-1]
    with pytest.raises(frieda_amd.FriedaPanic):
        frieda_amd.verify(p, None)
    p = base.clone()
    e = p.evaluations
    e[[0, 1]] = e[[1, 0]]
    p.evaluations = e
    assert not frieda_amd.verify(p, None)


def test_tampered_witnesses_are_rejected(proofs):
    import frieda_amd

    img = bytearray(proofs["p1024"][1].serialize())
    base = frieda_amd.Proof.deserialize(bytes(img))
    assert frieda_amd.verify(base, 1024)
    rng = np.random.default_rng(1)
    rejected = 0
    trials = 0
    for _ in range(60):
        pos = int(rng.integers(36, len(img)))  # past the header words
        img2 = bytearray(img)
        img2[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            p = frieda_amd.Proof.deserialize(bytes(img2))
        except frieda_amd.FriedaError:
            continue
        trials += 1
        try:
            if not frieda_amd.verify(p, 1024):
                rejected += 1
        except frieda_amd.FriedaPanic:
            rejected += 1
    assert trials > 20 and rejected == trials


def test_product_and_oracle_verifiers_agree_on_random_tampering(oracle, proofs):
    import frieda_amd

    _, op = proofs["small"]
    base = frieda_amd.Proof.deserialize(op.serialize())
    rng = np.random.default_rng(5)
    for _ in range(25):
        o2 = op.clone()
        p2 = base.clone()
        ev = p2.evaluations
        i, c = int(rng.integers(0, ev.shape[0])), int(rng.integers(0, 4))
        ev[i, c] = (int(ev[i, c]) + int(rng.integers(0, 2))) % P  # sometimes a no-op
        p2.evaluations = ev
        for k, v in enumerate(ev.ravel()):
            o2.c.evaluations[k] = int(v)
        assert frieda_amd.verify(p2, 9) == oracle.verify(o2, 9)


def test_rust_bindings_declare_every_symbol():
    """bindings/rust/frieda-hip-sys cannot be compiled here (no cargo); keep it in lock-step with the header: same function
    names, same number of parameters."""
    import os
    import re

    from frieda_amd import _lib

    hdr = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER_PATH).read() + open(_lib.TESTING_HEADER_PATH).read(), flags=re.S)
    decl = {}
    for m in re.finditer(r"\b(frieda_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr):
        args = m.group(2).strip()
        decl[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    rs = open(os.path.join(os.path.dirname(_lib.HEADER_PATH), "..", "bindings", "rust", "frieda-hip-sys", "src", "lib.rs")).read()
    rdecl = {}
    for m in re.finditer(r"pub fn (frieda_[a-z0-9_]+)\s*\(([^)]*)\)", rs):
        args = m.group(2).strip()
        rdecl[m.group(1)] = 0 if args == "" else args.count(":")
    assert sorted(decl) == sorted(rdecl)
    assert {k: v for k, v in decl.items() if rdecl[k] != v} == {}


def test_opening_tables_equal_the_reference_decommit_walk():
    """decommit.hip builds every opening list of a proof from one family of tables: with uq the sorted unique queries,
    U_s = unique(uq >> s) and E_s = the children of U_s's nodes that are missing from U_{s-1}.  The claim (frieda_amd/csrc/
    decommit.hip header): FRI layer li's witness positions are E_{li+1} and its Merkle hash witness is, bottom-up, the nodes
    E_{li+2} .. E_n at tree level n - s + 1.  Checked here against a literal transcription of the reference's walk
    (stwo core/fri.rs compute_decommitment_positions_and_witness_evals + core/vcs/prover.rs MerkleProver::decommit)."""
    import random

    def tables(uq, n):
        E = {}
        for s in range(1, n + 1):
            prev = set(q >> (s - 1) for q in uq)
            out = []
            for v in sorted(set(q >> s for q in uq)):
                for child in (2 * v, 2 * v + 1):
                    if child not in prev:
                        out.append(child)
            E[s] = out
        return E

    def reference_layer(queries, log):
        # positions + witness (fold_step = 1)
        pos, witness = [], []
        for pair in sorted(set(q >> 1 for q in queries)):
            for p in (2 * pair, 2 * pair + 1):
                pos.append(p)
                if p not in queries:
                    witness.append(p)
        # Merkle decommit: layers from the leaves up; emit the hash of every child that was not visited
        hashes = []  # (level, node)
        below = pos
        for layer in range(log - 1, -1, -1):
            here = []
            i = 0
            while i < len(below):
                node = below[i] >> 1
                has_left = below[i] == 2 * node
                if has_left:
                    i += 1
                has_right = i < len(below) and below[i] == 2 * node + 1
                if has_right:
                    i += 1
                if not has_left:
                    hashes.append((layer + 1, 2 * node))
                if not has_right:
                    hashes.append((layer + 1, 2 * node + 1))
                here.append(node)
            below = here
        return witness, hashes

    rng = random.Random(5)
    for _ in range(200):
        n = rng.randint(2, 14)
        nq = rng.choice([1, 2, 3, 7, 20, 64, 200])
        uq = sorted(set(rng.randrange(1 << n) for _ in range(nq)))
        n_layers = rng.randint(1, n - 1)
        E = tables(uq, n)
        for li in range(n_layers):
            lq = set(q >> li for q in uq)
            witness, hashes = reference_layer(lq, n - li)
            assert witness == E[li + 1], (n, li)
            mine = [(n - s + 1, c) for s in range(li + 2, n + 1) for c in E[s]]
            assert mine == hashes, (n, li)


def test_traffic_fingerprint_is_one_function():
    """bench.py reports `roofline.traffic` from a committed PMC pass and says whether frieda_amd/csrc changed since: the fingerprint in the
    traffic file (tools/traffic_from_pmc.py) and the one bench.py computes must be the same function of the tree."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    b = load("bench_mod", os.path.join(root, "bench.py"))
    t = load("traffic_mod", os.path.join(root, "tools", "traffic_from_pmc.py"))
    assert b._csrc_sha16() == t.csrc_sha16() and len(b._csrc_sha16()) == 16
    val, src, per_blob = b.traffic_from_profiles("tree5_leaf", 24, "prove")  # (the dominant launch of a proof since round 5)
    assert val and val > 1e8 and "collected at commit" in src and ("unchanged since" in src or "STALE" in src)
    assert b.traffic_from_profiles("tree5_leaf", 22, "prove")[0] is None
