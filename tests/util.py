"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

from conftest import GOLDEN, pattern_bytes


def load_vectors():
    with open(os.path.join(GOLDEN, "vectors.json")) as f:
        return json.load(f)


def resolve_input(spec, blob):
    if spec == "blob":
        return blob
    kind, arg = spec.split(":", 1)
    if kind == "pattern":
        return pattern_bytes(int(arg)).tobytes()
    if kind == "ascii":
        return arg.encode()
    raise ValueError(spec)


def blob_len_for(log_domain, log_blowup=4):
    """Byte length whose felts exactly fill a 2^log_domain domain: 4*2^L felts of 30 bits."""
    L = log_domain - log_blowup
    return (4 << L) * 30 // 8


class DevBuf:
    """Device memory through the C ABI's Column storage (frieda_dev_alloc / upload / download)."""

    def __init__(self, ctx, nbytes):
        import ctypes as C

        self.ctx = ctx
        self.nbytes = nbytes
        self.ptr = C.c_void_p()
        from frieda_amd.api import _check

        _check(ctx._L.frieda_dev_alloc(ctx._h, nbytes, C.byref(self.ptr)), ctx._h)

    @classmethod
    def from_array(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(ctx, max(arr.nbytes, 1))
        from frieda_amd.api import _check

        if arr.nbytes:
            _check(ctx._L.frieda_dev_upload(ctx._h, b.ptr, arr.ctypes.data, arr.nbytes), ctx._h)
        return b

    def to_array(self, dtype, shape):
        from frieda_amd.api import _check

        out = np.zeros(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        if out.nbytes:
            _check(self.ctx._L.frieda_dev_download(self.ctx._h, out.ctypes.data, self.ptr, out.nbytes), self.ctx._h)
        return out

    def free(self):
        if self.ptr:
            self.ctx._L.frieda_dev_free(self.ctx._h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
