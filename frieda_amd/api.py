"""Host-side mirror of frieda's public API over the C ABI (include/frieda_hip.h).

Same names, argument meaning and error behaviour as /root/reference/src/lib.rs:22-44:

    commit(data, log_blowup_factor) -> bytes[32]
    generate_proof(data, seed, pcs_config) -> Proof
    commit_and_generate_proof(data, seed, pcs_config) -> (bytes[32], Proof)      (src/proof.rs:32)
    verify(proof, seed) -> bool

Where the reference panics (assert!/unwrap, e.g. src/proof.rs:166-173) `FriedaPanic` is raised; verifier
rejections return False.  Every call runs on the MI355X through libfrieda_hip.so; there is no CPU path.
"""
import ctypes as C
import threading
import time
from dataclasses import dataclass

import numpy as np

from . import _lib


class FriedaError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = _lib.lib().frieda_status_string(status).decode()
        super().__init__(f"frieda_hip status {status} ({msg}){': ' + detail if detail else ''}")


class FriedaPanic(FriedaError):
    """The reference implementation panics on this input (FRIEDA_ERR_INVARIANT)."""


def _check(status, ctx=None):
    if status == _lib.OK:
        return
    detail = ""
    if ctx is not None:
        detail = _lib.lib().frieda_last_error(ctx).decode(errors="replace")
    raise (FriedaPanic if status == _lib.ERR_INVARIANT else FriedaError)(status, detail)


@dataclass(frozen=True)
class FriConfig:
    """stwo FriConfig as constructed at src/proof.rs:110-114."""

    log_blowup_factor: int = 4
    log_last_layer_degree_bound: int = 0
    n_queries: int = 20


@dataclass(frozen=True)
class PcsConfig:
    """stwo PcsConfig as constructed at src/proof.rs:109-116, benches/proof.rs:5-12."""

    fri_config: FriConfig = FriConfig()
    pow_bits: int = 20

    def _c(self):
        f = self.fri_config
        return _lib.PcsConfigC(self.pow_bits, f.log_blowup_factor, f.log_last_layer_degree_bound, f.n_queries)


def _seed_ptr(seed):
    return C.byref(C.c_uint64(seed)) if seed is not None else None


def _as_bytes(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


class Context:
    """One device + one stream + twiddle cache + workspace (frieda_ctx).  Not thread-safe; use one per GPU/thread."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        self._L = _lib.lib()
        _check(self._L.frieda_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            self._L.frieda_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(self._L.frieda_ctx_synchronize(self._h), self._h)

    def release_workspace(self):
        """Free the device workspace, the pinned staging block and the twiddle tables (they are re-created on demand)."""
        _check(self._L.frieda_ctx_release_workspace(self._h), self._h)

    def set_twiddle_cache(self, enabled):
        _check(self._L.frieda_ctx_set_twiddle_cache(self._h, int(bool(enabled))), self._h)

    def set_option(self, name, value):
        """A tuning / A-B option of this context alone (frieda_ctx_set_option): `name` is the environment variable that sets its default
        at context creation, e.g. "FRIEDA_HOST_DECOMMIT"; every option selects another kernel or plan for the same result."""
        _check(self._L.frieda_ctx_set_option(self._h, name.encode(), int(value)), self._h)

    def set_host_channel(self, enabled):
        """Evaluate the Fiat-Shamir channel on the host between layers instead of inside the device kernels."""
        _check(self._L.frieda_ctx_set_host_channel(self._h, int(bool(enabled))), self._h)

    def set_kernel_timing(self, enabled):
        _check(self._L.frieda_ctx_set_kernel_timing(self._h, int(bool(enabled))), self._h)

    def last_prove_phases(self):
        """Host wall-clock marks (ms since entry) of the last generate_proof: enqueued, device done, queries, gather, assembled."""
        a = (C.c_double * 8)()
        _check(self._L.frieda_ctx_last_prove_phases(self._h, a), self._h)
        return dict(zip(["enqueued", "device_done", "queries", "gathered", "assembled", "first_launch_after_entry"], list(a)[:6]))

    def blake2s_ceiling(self):
        """(leaf, node) compressions per second this device sustains right now on register-resident data (measurement aid)."""
        a, b = C.c_double(), C.c_double()
        _check(self._L.frieda_ctx_blake2s_ceiling(self._h, C.byref(a), C.byref(b)), self._h)
        return a.value, b.value

    def blake2s_ceiling_ex(self):
        """The ceiling with the in-kernel clock: {leaf_per_s, node_per_s, leaf_clock_ghz, leaf_cycles, node_clock_ghz, node_cycles}
        (cycles = SIMD cycles per wave-compression)."""
        a = (C.c_double * 6)()
        _check(self._L.frieda_ctx_blake2s_ceiling_ex(self._h, a), self._h)
        return dict(zip(["leaf_per_s", "node_per_s", "leaf_clock_ghz", "leaf_cycles_per_wave_compression", "node_clock_ghz", "node_cycles_per_wave_compression"], list(a)))

    def last_transcript(self):
        """Diagnostic: alphas drawn per FRI layer and the channel digest the grind was keyed by, of the last finished proof."""
        n = C.c_uint32()
        alphas = (C.c_uint32 * (4 * 64))()
        digest = (C.c_uint8 * 32)()
        _check(self._L.frieda_ctx_last_transcript(self._h, C.byref(n), alphas, 64, digest), self._h)
        return {"alphas": [[int(alphas[4 * i + c]) for c in range(4)] for i in range(min(n.value, 64))], "digest_before_grind": bytes(digest)}

    def kernel_timing_report(self, reset=True):
        """Per-kernel HIP-event timings accumulated since the last reset: list of dicts."""
        import json

        n = self._L.frieda_ctx_kernel_timing_report(self._h, None, 0, 0)
        buf = C.create_string_buffer(n + 16)
        self._L.frieda_ctx_kernel_timing_report(self._h, buf, n + 16, int(bool(reset)))
        return json.loads(buf.value.decode())["kernels"]

    # ---- Level A ----
    def commit(self, data, log_blowup_factor):
        a = _as_bytes(data)
        root = (C.c_uint8 * 32)()
        _check(self._L.frieda_commit(self._h, a.ctypes.data if a.size else None, a.size, log_blowup_factor, root), self._h)
        return bytes(root)

    def commit_device(self, d_ptr, length, log_blowup_factor, d_root_ptr):
        """Blob and 32-byte root both in device memory; asynchronous on the context stream."""
        _check(self._L.frieda_commit_device(self._h, d_ptr, length, log_blowup_factor, d_root_ptr), self._h)

    # ---- reconstruction side (host arrays in, bytes out; the device entry points are frieda_reconstruct_*_device) ----
    def _dev_call_with_upload(self, host_arr, n_out_bytes, call):
        """uploads a uint32 array, runs call(d_in, d_out), downloads n_out_bytes"""
        import numpy as np

        a = np.ascontiguousarray(host_arr, dtype=np.uint32)
        d_in, d_out = C.c_void_p(), C.c_void_p()
        _check(self._L.frieda_dev_alloc(self._h, a.nbytes, C.byref(d_in)), self._h)
        try:
            _check(self._L.frieda_dev_alloc(self._h, n_out_bytes + 16, C.byref(d_out)), self._h)
            _check(self._L.frieda_dev_upload(self._h, d_in, a.ctypes.data, a.nbytes), self._h)
            call(d_in, d_out)
            out = (C.c_uint8 * max(n_out_bytes, 1))()
            if n_out_bytes:
                _check(self._L.frieda_dev_download(self._h, out, d_out, n_out_bytes), self._h)
            return bytes(out)[:n_out_bytes]
        finally:
            self._L.frieda_dev_free(self._h, d_in)
            if d_out:
                self._L.frieda_dev_free(self._h, d_out)

    def reconstruct_from_block(self, block, log_domain, block_index, n_bytes):
        """block: uint32[4, 2^L] = entries block_index * 2^L .. of the four bit-reversed evaluation columns -> the blob."""
        L = int(block.shape[1]).bit_length() - 1
        return self._dev_call_with_upload(
            block, n_bytes, lambda d_in, d_out: _check(self._L.frieda_reconstruct_device(self._h, d_in, L, log_domain, block_index, n_bytes, d_out), self._h)
        )

    def reconstruct_from_cells(self, cells, cell_index, log_coef, log_domain, n_bytes):
        """cells: uint32[R, 4, 2^m] (cell r = entries cell_index[r] * 2^m .. of every column), R = 2^(log_coef - m) <= 4096
        distinct cells anywhere in the codeword -> the blob."""
        import numpy as np

        R, m = int(cells.shape[0]), int(cells.shape[2]).bit_length() - 1
        idx = np.ascontiguousarray(cell_index, dtype=np.uint32)
        return self._dev_call_with_upload(
            cells,
            n_bytes,
            lambda d_in, d_out: _check(
                self._L.frieda_reconstruct_cells_device(self._h, d_in, idx.ctypes.data, R, m, log_coef, log_domain, n_bytes, d_out), self._h
            ),
        )

    def reconstruct_from_points(self, cells, cell_index, log_coef, log_domain, n_bytes):
        """cells: uint32[R, 4, 2^m] as in reconstruct_from_cells, but ANY number of distinct cells holding at least 2^log_coef + 2
        points in all (single sampled points: m = 0) — no linear system, no bound on R (frieda_reconstruct_points_device)."""
        import numpy as np

        R, m = int(cells.shape[0]), int(cells.shape[2]).bit_length() - 1
        idx = np.ascontiguousarray(cell_index, dtype=np.uint32)
        return self._dev_call_with_upload(
            cells,
            n_bytes,
            lambda d_in, d_out: _check(
                self._L.frieda_reconstruct_points_device(self._h, d_in, idx.ctypes.data, R, m, log_coef, log_domain, n_bytes, d_out), self._h
            ),
        )

    def commit_and_generate_proof(self, data, seed, pcs_config):
        a = _as_bytes(data)
        root = (C.c_uint8 * 32)()
        out = C.c_void_p()
        _check(
            self._L.frieda_commit_and_generate_proof(
                self._h, a.ctypes.data if a.size else None, a.size, _seed_ptr(seed), pcs_config._c(), root, C.byref(out)
            ),
            self._h,
        )
        return bytes(root), Proof(out)

    def commit_and_generate_proof_device(self, d_ptr, length, seed, pcs_config):
        root = (C.c_uint8 * 32)()
        out = C.c_void_p()
        _check(
            self._L.frieda_commit_and_generate_proof_device(self._h, d_ptr, length, _seed_ptr(seed), pcs_config._c(), root, C.byref(out)),
            self._h,
        )
        return bytes(root), Proof(out)

    def generate_proof(self, data, seed, pcs_config):
        return self.commit_and_generate_proof(data, seed, pcs_config)[1]

    # ---- split form: overlap several proofs on one GPU with one Context per in-flight proof ----
    def prove_begin(self, data, seed, pcs_config):
        a = _as_bytes(data)
        self._keep = a  # the host blob must outlive the asynchronous upload
        _check(self._L.frieda_prove_begin(self._h, a.ctypes.data if a.size else None, a.size, _seed_ptr(seed), pcs_config._c()), self._h)

    def prove_begin_device(self, d_ptr, length, seed, pcs_config):
        _check(self._L.frieda_prove_begin_device(self._h, d_ptr, length, _seed_ptr(seed), pcs_config._c()), self._h)

    # ---- batches of equal-length blobs: every kernel handles the whole batch (include/frieda_hip.h) ----
    def _seeds_array(self, seeds, count):
        if seeds is None:
            return None
        if len(seeds) != count:
            raise ValueError("one seed per blob")
        return (C.c_uint64 * count)(*[int(s) for s in seeds])

    def commit_and_generate_proof_batch(self, blobs, seeds, pcs_config):
        """blobs: equal-length bytes-like objects; seeds: None or one int per blob.  Returns [(commitment, Proof), ...]."""
        count = len(blobs)
        if count == 0:
            return []
        length = len(blobs[0])
        if any(len(b) != length for b in blobs):
            raise ValueError("a batch holds blobs of one length")
        flat = b"".join(bytes(b) for b in blobs)
        buf = (C.c_uint8 * max(len(flat), 1)).from_buffer_copy(flat or b"\0")
        return self._prove_batch(self._L.frieda_commit_and_generate_proof_batch, buf, length, length, count, seeds, pcs_config)

    def commit_and_generate_proof_batch_device(self, d_ptr, stride, length, count, seeds, pcs_config):
        return self._prove_batch(self._L.frieda_commit_and_generate_proof_batch_device, d_ptr, stride, length, count, seeds, pcs_config)

    def prove_batch_begin_device(self, d_ptr, stride, length, count, seeds, pcs_config):
        """Enqueue the commit phase of a batch; the blobs must stay valid until prove_batch_finish(count) returns."""
        _check(self._L.frieda_prove_batch_begin_device(self._h, d_ptr, stride, length, count, self._seeds_array(seeds, count), pcs_config._c()), self._h)

    def prove_batch_finish(self, count):
        roots = (C.c_uint8 * (32 * count))()
        outs = (C.c_void_p * count)()
        _check(self._L.frieda_prove_batch_finish(self._h, count, roots, outs), self._h)
        rb = bytes(roots)
        return [(rb[32 * i : 32 * i + 32], Proof(C.c_void_p(outs[i]))) for i in range(count)]

    def _prove_batch(self, fn, data, stride, length, count, seeds, pcs_config):
        roots = (C.c_uint8 * (32 * count))()
        outs = (C.c_void_p * count)()
        _check(fn(self._h, data, stride, length, count, self._seeds_array(seeds, count), pcs_config._c(), roots, outs), self._h)
        rb = bytes(roots)
        return [(rb[32 * i : 32 * i + 32], Proof(C.c_void_p(outs[i]))) for i in range(count)]

    def commit_batch(self, blobs, log_blowup_factor):
        count = len(blobs)
        if count == 0:
            return []
        length = len(blobs[0])
        if any(len(b) != length for b in blobs):
            raise ValueError("a batch holds blobs of one length")
        flat = b"".join(bytes(b) for b in blobs)
        buf = (C.c_uint8 * max(len(flat), 1)).from_buffer_copy(flat or b"\0")
        roots = (C.c_uint8 * (32 * count))()
        _check(self._L.frieda_commit_batch(self._h, buf, length, length, count, log_blowup_factor, roots), self._h)
        rb = bytes(roots)
        return [rb[32 * i : 32 * i + 32] for i in range(count)]

    def commit_batch_device(self, d_ptr, stride, length, count, log_blowup_factor):
        roots = (C.c_uint8 * (32 * count))()
        _check(self._L.frieda_commit_batch_device(self._h, d_ptr, stride, length, count, log_blowup_factor, roots), self._h)
        rb = bytes(roots)
        return [rb[32 * i : 32 * i + 32] for i in range(count)]

    def prove_finish(self):
        root = (C.c_uint8 * 32)()
        out = C.c_void_p()
        _check(self._L.frieda_prove_finish(self._h, root, C.byref(out)), self._h)
        self._keep = None
        return bytes(root), Proof(out)


class Proof:
    """frieda::proof::Proof (src/proof.rs:19-26).  Fields are public upstream, so they are readable and writable here."""

    def __init__(self, handle):
        self._h = handle
        self._L = _lib.lib()

    def __del__(self):
        try:
            if self._h:
                self._L.frieda_proof_free(self._h)
                self._h = None
        except Exception:
            pass

    def clone(self):
        out = C.c_void_p()
        _check(self._L.frieda_proof_clone(self._h, C.byref(out)))
        return Proof(out)

    @property
    def proof_of_work(self):
        return self._L.frieda_proof_proof_of_work(self._h)

    @proof_of_work.setter
    def proof_of_work(self, v):
        self._L.frieda_proof_set_proof_of_work(self._h, v)

    @property
    def pcs_config(self):
        c = self._L.frieda_proof_pcs_config(self._h)
        return PcsConfig(FriConfig(c.log_blowup_factor, c.log_last_layer_degree_bound, c.n_queries), c.pow_bits)

    @property
    def log_size_bound(self):
        return self._L.frieda_proof_log_size_bound(self._h)

    @property
    def evaluations(self):
        """uint32[n, 4] copy of Vec<QM31>."""
        n = self._L.frieda_proof_n_evaluations(self._h)
        if n == 0:
            return np.zeros((0, 4), dtype=np.uint32)
        p = self._L.frieda_proof_evaluations(self._h)
        return np.ctypeslib.as_array(p, shape=(n, 4)).copy()

    @evaluations.setter
    def evaluations(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint32).reshape(-1, 4)
        _check(self._L.frieda_proof_resize_evaluations(self._h, arr.shape[0]))
        if arr.shape[0]:
            p = self._L.frieda_proof_evaluations(self._h)
            C.memmove(p, arr.ctypes.data, arr.nbytes)

    @property
    def n_inner_layers(self):
        return self._L.frieda_proof_n_inner_layers(self._h)

    def layer(self, i):
        """Layer 0 = first_layer, 1.. = inner_layers[i-1] -> dict(commitment, fri_witness, hash_witness, column_witness)."""
        n = C.c_size_t()
        com = bytes(self._L.frieda_proof_layer_commitment(self._h, i)[:32])
        p = self._L.frieda_proof_layer_fri_witness(self._h, i, C.byref(n))
        fw = np.ctypeslib.as_array(p, shape=(n.value, 4)).copy() if n.value else np.zeros((0, 4), np.uint32)
        p = self._L.frieda_proof_layer_hash_witness(self._h, i, C.byref(n))
        hw = [bytes(p[32 * j : 32 * j + 32]) for j in range(n.value)]
        p = self._L.frieda_proof_layer_column_witness(self._h, i, C.byref(n))
        cw = np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros((0,), np.uint32)
        return {"commitment": com, "fri_witness": fw, "hash_witness": hw, "column_witness": cw}

    @property
    def commitment(self):
        return bytes(self._L.frieda_proof_layer_commitment(self._h, 0)[:32])

    @property
    def last_layer_poly(self):
        n = C.c_size_t()
        p = self._L.frieda_proof_last_layer_poly(self._h, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value, 4)).copy() if n.value else np.zeros((0, 4), np.uint32)

    def serialize(self):
        n = self._L.frieda_proof_serialize(self._h, None, 0)
        buf = (C.c_uint8 * n)()
        self._L.frieda_proof_serialize(self._h, buf, n)
        return bytes(buf)

    @staticmethod
    def deserialize(blob):
        a = np.frombuffer(bytes(blob), dtype=np.uint8)
        out = C.c_void_p()
        _check(_lib.lib().frieda_proof_deserialize(a.ctypes.data, a.size, C.byref(out)))
        return Proof(out)


class MultiContext:
    """A batch of independent blobs across the GPUs of one node from ONE process (frieda_multi, include/frieda_hip.h): blob i runs
    on devices[i mod n] (one host thread + two contexts per device), the 32-byte roots are gathered with ncclAllGather on a
    single-process RCCL communicator.  The one-process-per-GPU launch model (bench.py, torch.distributed) lives in batch.py."""

    def __init__(self, devices):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        st = self._L.frieda_multi_create(devs, len(devices), C.byref(self._h))
        if st != _lib.OK:
            raise FriedaError(st, f"frieda_multi_create({list(devices)})")

    def _check(self, st):
        if st != _lib.OK:
            detail = self._L.frieda_multi_last_error(self._h).decode(errors="replace")
            raise (FriedaPanic if st == _lib.ERR_INVARIANT else FriedaError)(st, detail)

    def close(self):
        if self._h:
            self._L.frieda_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def release_workspace(self):
        """Frees the device workspaces (two per device, up to the batch budget each) and upload rings the handle keeps between calls."""
        self._check(self._L.frieda_multi_release_workspace(self._h))

    @property
    def device_count(self):
        return self._L.frieda_multi_device_count(self._h)

    @property
    def uses_rccl(self):
        return bool(self._L.frieda_multi_uses_rccl(self._h))

    @property
    def gather_count(self):
        return int(self._L.frieda_multi_gather_count(self._h))

    def _blob_table(self, blobs):
        arrs = [_as_bytes(b) for b in blobs]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data if a.size else None for a in arrs])
        lens = (C.c_size_t * len(arrs))(*[a.size for a in arrs])
        return arrs, ptrs, lens

    def commit_many(self, blobs, log_blowup_factor):
        if not blobs:
            return []
        arrs, ptrs, lens = self._blob_table(blobs)
        roots = (C.c_uint8 * (32 * len(arrs)))()
        self._check(self._L.frieda_commit_many(self._h, ptrs, lens, len(arrs), log_blowup_factor, roots))
        rb = bytes(roots)
        return [rb[32 * i : 32 * i + 32] for i in range(len(arrs))]

    def prove_many(self, blobs, seeds, pcs_config):
        """-> [(commitment, Proof)] in blob order; seeds: None or one per blob"""
        if not blobs:
            return []
        arrs, ptrs, lens = self._blob_table(blobs)
        n = len(arrs)
        if seeds is not None and len(seeds) != n:
            raise ValueError("one seed per blob")
        sd = (C.c_uint64 * n)(*seeds) if seeds is not None else None
        roots = (C.c_uint8 * (32 * n))()
        outs = (C.c_void_p * n)()
        self._check(self._L.frieda_prove_many(self._h, ptrs, lens, n, sd, pcs_config._c(), roots, outs))
        rb = bytes(roots)
        return [(rb[32 * i : 32 * i + 32], Proof(C.c_void_p(outs[i]))) for i in range(n)]


class ProofPipeline:
    """Keeps up to `depth` proofs in flight on one GPU (one Context = stream + workspace each), so that the latency-bound
    tail of one proof (tree tops, small FRI layers, host round trips) overlaps the throughput-bound kernels of the next.
    submit() returns the oldest finished (commitment, proof) once the pipeline is full, else None; drain() flushes."""

    def __init__(self, device=0, depth=2):
        self.ctxs = [Context(device) for _ in range(depth)]
        self.inflight = []  # contexts with a proof in flight, oldest first
        self.free = list(self.ctxs)

    def submit_device(self, d_ptr, length, seed, pcs_config):
        done = None
        if not self.free:
            ctx = self.inflight.pop(0)
            done = ctx.prove_finish()
            self.free.append(ctx)
        ctx = self.free.pop(0)
        ctx.prove_begin_device(d_ptr, length, seed, pcs_config)
        self.inflight.append(ctx)
        return done

    def drain(self):
        out = []
        while self.inflight:
            ctx = self.inflight.pop(0)
            out.append(ctx.prove_finish())
            self.free.append(ctx)
        return out

    def close(self):
        self.drain()
        for c in self.ctxs:
            c.close()


class BatchPipeline:
    """ProofPipeline for batches: up to `depth` batches of equal-length blobs in flight on one GPU (frieda_prove_batch_begin_device /
    _finish on one Context each).  The Fiat-Shamir chain is paid once per batch AND runs under the other batch's wide kernels —
    the highest-throughput way through a stream of blobs.  submit returns the oldest finished batch's [(commitment, proof)] or None."""

    def __init__(self, device=0, depth=2):
        self.ctxs = [Context(device) for _ in range(depth)]
        self.inflight = []  # (ctx, count, time of _begin), oldest first
        self.free = list(self.ctxs)
        self.call_latencies = []  # (blobs, seconds from _begin to the return of _finish) of every finished call; callers may clear it

    def plan(self, length, count, pcs_config, prove=True):
        """The library's cut of `count` equal-length blobs into calls for this pipeline's depth (frieda_batch_plan: workspace bytes in
        flight; the options FRIEDA_BATCH_BUDGET_MB / FRIEDA_BATCH_CALLS_PER_CTX of the first context apply)."""
        return batch_plan(length, count, pcs_config, in_flight=len(self.ctxs), prove=prove, ctx=self.ctxs[0])

    def run_stream_device(self, d_ptr, stride, length, count, seeds, pcs_config):
        """`count` device-resident blobs of one length (blob i at d_ptr + i * stride) through the pipeline, cut into calls by the library's
        batch policy; returns [(commitment, proof)] in blob order (the caller loop of benches/proof.rs:30-44 over many blobs)."""
        out = []
        i = 0
        for cnt in self.plan(length, count, pcs_config):
            sd = None if seeds is None else seeds[i : i + cnt]
            r = self.submit_device(d_ptr + i * stride, stride, length, cnt, sd, pcs_config)
            if r is not None:
                out.extend(r)
            i += cnt
        out.extend(self.drain())
        return out

    def submit_device(self, d_ptr, stride, length, count, seeds, pcs_config):
        done = None
        if not self.free:
            done = self._finish_oldest()
        ctx = self.free.pop(0)
        t0 = time.perf_counter()
        ctx.prove_batch_begin_device(d_ptr, stride, length, count, seeds, pcs_config)
        self.inflight.append((ctx, count, t0))
        return done

    def _finish_oldest(self):
        ctx, cnt, t0 = self.inflight.pop(0)
        done = ctx.prove_batch_finish(cnt)
        self.call_latencies.append((cnt, time.perf_counter() - t0))
        self.free.append(ctx)
        return done

    def drain(self):
        out = []
        while self.inflight:
            out.extend(self._finish_oldest())
        return out

    def close(self):
        self.drain()
        for c in self.ctxs:
            c.close()


def workspace_bytes(length, log_blowup_factor, log_last_layer_degree_bound=0, prove=True):
    """Device workspace one blob of `length` bytes adds to a batched call (frieda_workspace_bytes); 0: shape out of range."""
    return int(_lib.lib().frieda_workspace_bytes(length, log_blowup_factor, log_last_layer_degree_bound, int(bool(prove))))


def batch_plan(length, count, pcs_config=None, in_flight=2, prove=True, ctx=None, log_blowup_factor=None):
    """The library's batch policy (frieda_batch_plan, include/frieda_hip.h "batch policy"): the blob count of each call, in order, for
    `count` equal-length blobs and `in_flight` contexts taking turns.  pcs_config for proofs, log_blowup_factor for commits."""
    L = _lib.lib()
    if pcs_config is not None:
        B, last = pcs_config.fri_config.log_blowup_factor, pcs_config.fri_config.log_last_layer_degree_bound
    else:
        B, last = int(log_blowup_factor), 0
    n = C.c_uint32(0)
    h = ctx._h if ctx is not None else None
    _check(L.frieda_batch_plan(h, length, B, last, int(bool(prove)), count, in_flight, None, 0, C.byref(n)))
    out = (C.c_uint32 * max(1, n.value))()
    _check(L.frieda_batch_plan(h, length, B, last, int(bool(prove)), count, in_flight, out, n.value, C.byref(n)))
    return [int(out[i]) for i in range(n.value)]


# ---- module-level API with an implicit per-thread default context (frieda's free functions) ----
_tls = threading.local()


def _default_device():
    """The GPU of this process: torch's current device when torch has initialised one (one process per GPU under
    torch.distributed), else LOCAL_RANK, else 0."""
    import os
    import sys

    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        return int(torch.cuda.current_device())
    return int(os.environ.get("LOCAL_RANK", "0"))


def default_context():
    ctx = getattr(_tls, "ctx", None)
    if ctx is None:
        ctx = _tls.ctx = Context(_default_device())
    return ctx


def commit(data, log_blowup_factor):
    """api::commit (src/lib.rs:31)."""
    return default_context().commit(data, log_blowup_factor)


def generate_proof(data, seed, pcs_config):
    """api::generate_proof (src/lib.rs:36)."""
    return default_context().generate_proof(data, seed, pcs_config)


def commit_and_generate_proof(data, seed, pcs_config):
    """proof::commit_and_generate_proof (src/proof.rs:32)."""
    return default_context().commit_and_generate_proof(data, seed, pcs_config)


def verify(proof, seed):
    """api::verify (src/lib.rs:41): bool; raises FriedaPanic where the reference panics.  Host-only."""
    ok = C.c_int(0)
    _check(_lib.lib().frieda_verify(proof._h, _seed_ptr(seed), C.byref(ok)))
    return bool(ok.value)


def verify_samples(proof, seed):
    """verify + where the accepted proof sampled: (ok, positions) with positions[i] the index in the bit-reversed codeword whose four
    column values are proof.evaluations[i] (frieda_verify_samples); positions is None when the proof is rejected."""
    import numpy as np

    ok = C.c_int(0)
    n = C.c_size_t(0)
    cap = max(1, int(proof.pcs_config.fri_config.n_queries))
    buf = np.zeros(cap, dtype=np.uint32)
    _check(_lib.lib().frieda_verify_samples(proof._h, _seed_ptr(seed), C.byref(ok), buf.ctypes.data, cap, C.byref(n)))
    if not ok.value:
        return False, None
    return True, buf[: n.value].copy()
