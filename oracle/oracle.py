"""ctypes binding of the CPU oracle (oracle/libfrieda_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (frieda_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfrieda_oracle.so")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


class PcsConfig(C.Structure):
    _fields_ = [
        ("pow_bits", C.c_uint32),
        ("log_blowup_factor", C.c_uint32),
        ("log_last_layer_degree_bound", C.c_uint32),
        ("n_queries", C.c_uint32),
    ]


class LayerProof(C.Structure):
    _fields_ = [
        ("fri_witness", u32p),
        ("n_fri_witness", C.c_size_t),
        ("hash_witness", u8p),
        ("n_hash_witness", C.c_size_t),
        ("column_witness", u32p),
        ("n_column_witness", C.c_size_t),
        ("commitment", C.c_uint8 * 32),
    ]


class Proof(C.Structure):
    _fields_ = [
        ("first_layer", LayerProof),
        ("inner_layers", C.POINTER(LayerProof)),
        ("n_inner_layers", C.c_size_t),
        ("last_layer_poly", u32p),
        ("n_last_layer_poly", C.c_size_t),
        ("proof_of_work", C.c_uint64),
        ("pcs_config", PcsConfig),
        ("log_size_bound", C.c_uint32),
        ("evaluations", u32p),
        ("n_evaluations", C.c_size_t),
    ]


class Channel(C.Structure):
    _fields_ = [("digest", C.c_uint8 * 32), ("n_challenges", C.c_uint64), ("n_sent", C.c_uint64)]


class Trace(C.Structure):
    _fields_ = [
        ("n_layers", C.c_uint32),
        ("alphas", (C.c_uint32 * 4) * 64),
        ("roots", (C.c_uint8 * 32) * 64),
        ("digest_before_grind", C.c_uint8 * 32),
    ]


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("frieda_oracle.c", "frieda_oracle.h")]
    if not force and os.path.exists(_LIB_PATH) and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "libfrieda_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.fo_felt_count.restype = C.c_size_t
        L.fo_felt_count.argtypes = [C.c_size_t]
        L.fo_padded_len.restype = C.c_size_t
        L.fo_padded_len.argtypes = [C.c_size_t]
        L.fo_bytes_to_felt_le.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.fo_polynomial_from_bytes.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, u32p]
        L.fo_precompute_twiddles.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
        L.fo_circle_evaluate.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.fo_circle_interpolate_block.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.fo_felts_to_bytes.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.fo_reconstruct_cells.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fo_reconstruct_cells.restype = C.c_int
        L.fo_reconstruct_points.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.fo_reconstruct_points.restype = C.c_int
        L.fo_merkle_commit_layer.argtypes = [C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p]
        L.fo_merkle_commit.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_void_p]
        L.fo_merkle_layer_offset.restype = C.c_size_t
        L.fo_merkle_layer_offset.argtypes = [C.c_uint32, C.c_uint32]
        L.fo_fold_circle_into_line.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p]
        L.fo_fold_line.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]
        L.fo_circle_extend.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.fo_circle_eval_at_point.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fo_fri_decompose.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_void_p), C.c_void_p]
        L.fo_blake2s256.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.fo_blake2s_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.fo_channel_init.argtypes = [C.POINTER(Channel)]
        L.fo_channel_mix_u64.argtypes = [C.POINTER(Channel), C.c_uint64]
        L.fo_channel_mix_root.argtypes = [C.POINTER(Channel), C.c_void_p]
        L.fo_channel_mix_felts.argtypes = [C.POINTER(Channel), C.c_void_p, C.c_size_t]
        L.fo_channel_draw_random_bytes.argtypes = [C.POINTER(Channel), C.c_void_p]
        L.fo_channel_draw_felt.argtypes = [C.POINTER(Channel), C.c_void_p]
        L.fo_test_set_draw_bound.argtypes = [C.c_uint32]
        L.fo_channel_trailing_zeros.restype = C.c_uint32
        L.fo_channel_trailing_zeros.argtypes = [C.POINTER(Channel)]
        L.fo_grind.restype = C.c_uint64
        L.fo_grind.argtypes = [C.POINTER(Channel), C.c_uint32]
        L.fo_queries_generate.restype = C.c_size_t
        L.fo_queries_generate.argtypes = [C.POINTER(Channel), C.c_uint32, C.c_size_t, C.c_void_p]
        L.fo_commit.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
        L.fo_commit_and_generate_proof.argtypes = [C.c_void_p, C.c_size_t, u64p, PcsConfig, C.c_void_p, C.POINTER(C.POINTER(Proof))]
        L.fo_verify.argtypes = [C.POINTER(Proof), u64p, C.POINTER(C.c_int)]
        L.fo_proof_free.argtypes = [C.POINTER(Proof)]
        L.fo_proof_clone.restype = C.POINTER(Proof)
        L.fo_proof_clone.argtypes = [C.POINTER(Proof)]
        L.fo_proof_serialize.restype = C.c_size_t
        L.fo_proof_serialize.argtypes = [C.POINTER(Proof), C.c_void_p, C.c_size_t]
        L.fo_last_trace.restype = C.POINTER(Trace)
        L.fo_point_from_index.argtypes = [C.c_uint32, u32p, u32p]
        L.fo_circle_domain_at.argtypes = [C.c_uint32, C.c_uint32, u32p, u32p]
        L.fo_bit_reverse_index.restype = C.c_uint32
        L.fo_bit_reverse_index.argtypes = [C.c_uint32, C.c_uint32]
        L.fo_qm31_mul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        for f in ("fo_m31_add", "fo_m31_sub", "fo_m31_mul"):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [C.c_uint32, C.c_uint32]
        L.fo_m31_inv.restype = C.c_uint32
        L.fo_m31_inv.argtypes = [C.c_uint32]
        _lib = L
    return _lib


def _buf(data):
    """bytes-like -> (ctypes pointer-compatible object, length); keeps a reference alive."""
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    return a, a.ctypes.data, a.size


def _ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


# ---- codec ---------------------------------------------------------------------------------------
def bytes_to_felt_le(data):
    a, p, n = _buf(data)
    out = np.zeros(lib().fo_felt_count(n), dtype=np.uint32)
    lib().fo_bytes_to_felt_le(p, n, out.ctypes.data)
    return out


def polynomial_from_bytes(data):
    """-> (coef[4, 2^L] uint32, L)"""
    a, p, n = _buf(data)
    fp = lib().fo_padded_len(lib().fo_felt_count(n))
    coef = np.zeros(fp, dtype=np.uint32)
    L = C.c_uint32()
    lib().fo_polynomial_from_bytes(p, n, coef.ctypes.data, C.byref(L))
    return coef.reshape(4, -1), L.value


def bit_reverse_column(col):
    """CpuBackend's ColumnOps::bit_reverse_column on a copy of `col` (1-D uint32, power-of-two length)."""
    v = np.ascontiguousarray(col, dtype=np.uint32).copy()
    lib().fo_bit_reverse_column(C.c_void_p(v.ctypes.data), C.c_uint32(v.size.bit_length() - 1))
    return v


# ---- NTT -----------------------------------------------------------------------------------------
def precompute_twiddles(n):
    half = max(1, 1 << (n - 1))
    tw = np.zeros(half, dtype=np.uint32)
    itw = np.zeros(half, dtype=np.uint32)
    lib().fo_precompute_twiddles(n, tw.ctypes.data, itw.ctypes.data)
    return tw, itw


def circle_evaluate(coef, n, tw=None):
    """coef: [k, 2^L] uint32 -> [k, 2^n] evaluations (bit-reversed order)."""
    coef = np.ascontiguousarray(coef, dtype=np.uint32)
    k, m = coef.shape
    L = m.bit_length() - 1
    if tw is None:
        tw, _ = precompute_twiddles(n)
    out = np.zeros((k, 1 << n), dtype=np.uint32)
    for c in range(k):
        lib().fo_circle_evaluate(coef[c].ctypes.data, L, n, tw.ctypes.data, out[c].ctypes.data)
    return out


def circle_interpolate_block(block, n, k, itw=None):
    """block [ncols, 2^L] = evaluations k*2^L .. (k+1)*2^L of the bit-reversed codeword -> coefficients [ncols, 2^L]."""
    block = np.ascontiguousarray(block, dtype=np.uint32)
    ncols, m = block.shape
    L = m.bit_length() - 1
    if itw is None:
        _, itw = precompute_twiddles(n)
    out = np.zeros_like(block)
    for c in range(ncols):
        lib().fo_circle_interpolate_block(block[c].ctypes.data, L, n, k, itw.ctypes.data, out[c].ctypes.data)
    return out


def reconstruct_cells(cells, cell_index, n, L, tw=None, itw=None):
    """cells [R, ncols, 2^m] (cell r = entries cell_index[r] * 2^m .. of every column of the bit-reversed codeword), R = 2^(L-m)
    distinct cells -> coefficients [ncols, 2^L]."""
    cells = np.ascontiguousarray(cells, dtype=np.uint32)
    R, ncols, M = cells.shape
    m = M.bit_length() - 1
    idx = np.ascontiguousarray(cell_index, dtype=np.uint32)
    if tw is None or itw is None:
        tw, itw = precompute_twiddles(n)
    out = np.zeros((ncols, 1 << L), dtype=np.uint32)
    for c in range(ncols):
        col = np.ascontiguousarray(cells[:, c, :])
        rc = lib().fo_reconstruct_cells(col.ctypes.data, idx.ctypes.data, R, m, L, n, tw.ctypes.data, itw.ctypes.data, out[c].ctypes.data)
        if rc != 0:
            raise ValueError("fo_reconstruct_cells: bad arguments")
    return out


def reconstruct_points(vals, positions, n, L):
    """vals [n_pts, ncols] = the columns' values at the bit-reversed positions `positions` (any >= 2^L + 2 distinct ones) ->
    coefficients [ncols, 2^L] (erasure-locator route; quadratic in the domain size: test sizes)."""
    vals = np.ascontiguousarray(vals, dtype=np.uint32)
    n_pts, ncols = vals.shape
    pos = np.ascontiguousarray(positions, dtype=np.uint32)
    out = np.zeros((ncols, 1 << L), dtype=np.uint32)
    for c in range(ncols):
        col = np.ascontiguousarray(vals[:, c])
        rc = lib().fo_reconstruct_points(col.ctypes.data, pos.ctypes.data, n_pts, L, n, out[c].ctypes.data)
        if rc == -2:
            raise ValueError("fo_reconstruct_points: the samples are not values of one polynomial")
        if rc != 0:
            raise ValueError("fo_reconstruct_points: bad arguments or too few distinct points")
    return out


def felts_to_bytes(felts, length):
    felts = np.ascontiguousarray(felts, dtype=np.uint32).ravel()
    out = np.zeros(max(length, 1), dtype=np.uint8)
    lib().fo_felts_to_bytes(felts.ctypes.data, felts.size, out.ctypes.data, length)
    return out[:length].tobytes()


# ---- Merkle --------------------------------------------------------------------------------------
def merkle_commit_layer(log_size, prev, cols):
    """prev: None or uint8[2^(log_size+1), 32]; cols: None or [k, 2^log_size] uint32 -> uint8[2^log_size, 32]"""
    out = np.zeros((1 << log_size, 32), dtype=np.uint8)
    if cols is not None:
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        ptrs = _ptr_array([cols[i] for i in range(cols.shape[0])])
        k = cols.shape[0]
    else:
        ptrs, k = None, 0
    pp = None
    if prev is not None:
        prev = np.ascontiguousarray(prev, dtype=np.uint8)
        pp = prev.ctypes.data
    lib().fo_merkle_commit_layer(log_size, pp, ptrs, k, out.ctypes.data)
    return out


def merkle_commit(cols):
    """cols [k, 2^m] -> list of layers, layers[l] = uint8[2^l, 32] (layers[0] = root)."""
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    k, sz = cols.shape
    m = sz.bit_length() - 1
    buf = np.zeros(32 * (2 * sz - 1), dtype=np.uint8)
    lib().fo_merkle_commit(_ptr_array([cols[i] for i in range(k)]), k, m, buf.ctypes.data)
    layers = []
    for l in range(m + 1):
        off = lib().fo_merkle_layer_offset(m, l)
        layers.append(buf[off : off + (32 << l)].reshape(-1, 32))
    return layers


# ---- folds ---------------------------------------------------------------------------------------
def fold_circle_into_line(src, alpha, dst=None):
    """src [4, N]; alpha [4]; dst [4, N/2] accumulated in place (zeros if None)."""
    src = np.ascontiguousarray(src, dtype=np.uint32)
    n = src.shape[1].bit_length() - 1
    if dst is None:
        dst = np.zeros((4, src.shape[1] // 2), dtype=np.uint32)
    alpha = np.ascontiguousarray(alpha, dtype=np.uint32)
    lib().fo_fold_circle_into_line(_ptr_array([dst[i] for i in range(4)]), _ptr_array([src[i] for i in range(4)]), n, alpha.ctypes.data)
    return dst


def fold_line(src, domain_n, alpha):
    """src [4, M] on the line domain of log size log2(M) derived from the circle domain of log size domain_n."""
    src = np.ascontiguousarray(src, dtype=np.uint32)
    m = src.shape[1].bit_length() - 1
    dst = np.zeros((4, src.shape[1] // 2), dtype=np.uint32)
    alpha = np.ascontiguousarray(alpha, dtype=np.uint32)
    lib().fo_fold_line(_ptr_array([src[i] for i in range(4)]), m, domain_n, alpha.ctypes.data, _ptr_array([dst[i] for i in range(4)]))
    return dst


# ---- trait methods frieda's path never calls (PolyOps::extend / eval_at_point, FriOps::decompose) ---------------
def circle_extend(coef, log_size):
    coef = np.ascontiguousarray(coef, dtype=np.uint32)
    out = np.empty(1 << log_size, dtype=np.uint32)
    lib().fo_circle_extend(coef.ctypes.data, coef.size.bit_length() - 1, log_size, out.ctypes.data)
    return out


def circle_eval_at_point(coef, px, py):
    """coef: 2^k M31 coefficients; (px, py): a circle point over QM31 (4 words each) -> QM31 value (4 words)"""
    coef = np.ascontiguousarray(coef, dtype=np.uint32)
    px = np.ascontiguousarray(px, dtype=np.uint32)
    py = np.ascontiguousarray(py, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().fo_circle_eval_at_point(coef.ctypes.data, coef.size.bit_length() - 1, px.ctypes.data, py.ctypes.data, out.ctypes.data)
    return out


def fri_decompose(ev):
    """ev [4, 2^k] (SoA QM31, bit-reversed) -> (g [4, 2^k], lambda [4])"""
    ev = np.ascontiguousarray(ev, dtype=np.uint32)
    g = np.zeros_like(ev)
    lam = np.zeros(4, dtype=np.uint32)
    lib().fo_fri_decompose(_ptr_array([ev[i] for i in range(4)]), ev.shape[1].bit_length() - 1, _ptr_array([g[i] for i in range(4)]), lam.ctypes.data)
    return g, lam


# ---- API -----------------------------------------------------------------------------------------
def commit(data, log_blowup_factor):
    a, p, n = _buf(data)
    root = (C.c_uint8 * 32)()
    rc = lib().fo_commit(p, n, log_blowup_factor, root)
    if rc != 0:
        raise RuntimeError(f"oracle commit: status {rc}")
    return bytes(root)


class OracleProof:
    def __init__(self, ptr, commitment=None):
        self.ptr = ptr
        self.commitment = commitment

    def __del__(self):
        if self.ptr:
            lib().fo_proof_free(self.ptr)
            self.ptr = None

    def clone(self):
        return OracleProof(lib().fo_proof_clone(self.ptr), self.commitment)

    @property
    def c(self):
        return self.ptr.contents

    def serialize(self):
        n = lib().fo_proof_serialize(self.ptr, None, 0)
        buf = (C.c_uint8 * n)()
        lib().fo_proof_serialize(self.ptr, buf, n)
        return bytes(buf)

    def evaluations(self):
        c = self.c
        return np.ctypeslib.as_array(c.evaluations, shape=(c.n_evaluations, 4)).copy()


def make_config(pow_bits=20, log_blowup_factor=4, log_last_layer_degree_bound=0, n_queries=20):
    return PcsConfig(pow_bits, log_blowup_factor, log_last_layer_degree_bound, n_queries)


def commit_and_generate_proof(data, seed, cfg):
    a, p, n = _buf(data)
    root = (C.c_uint8 * 32)()
    out = C.POINTER(Proof)()
    sp = C.byref(C.c_uint64(seed)) if seed is not None else None
    rc = lib().fo_commit_and_generate_proof(p, n, sp, cfg, root, C.byref(out))
    if rc != 0:
        raise RuntimeError(f"oracle prove: status {rc}")
    return bytes(root), OracleProof(out, bytes(root))


def verify(proof, seed):
    ok = C.c_int(0)
    sp = C.byref(C.c_uint64(seed)) if seed is not None else None
    rc = lib().fo_verify(proof.ptr, sp, C.byref(ok))
    if rc != 0:
        raise RuntimeError(f"oracle verify: status {rc} (reference panics here)")
    return bool(ok.value)


def last_trace():
    t = lib().fo_last_trace().contents
    nl = t.n_layers
    return {
        "alphas": np.array([list(t.alphas[i]) for i in range(nl)], dtype=np.uint32),
        "roots": [bytes(t.roots[i]) for i in range(nl)],
        "digest_before_grind": bytes(t.digest_before_grind),
    }
