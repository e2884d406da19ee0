"""ctypes loader of libfrieda_hip.so — the C ABI declared in include/frieda_hip.h.

There is no fallback: if the HIP library is missing or a symbol is absent the import of the binding fails loudly.
"""
import ctypes as C
import os
import re
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
# FRIEDA_HIP_LIB: an alternative build of the same library (A/B experiments: tools/build_variant.sh); never a fallback
LIB_PATH = os.environ.get("FRIEDA_HIP_LIB") or os.path.join(_PKG, "lib", "libfrieda_hip.so")
HEADER_PATH = os.path.join(_ROOT, "include", "frieda_hip.h")
TESTING_HEADER_PATH = os.path.join(_ROOT, "include", "frieda_hip_testing.h")  # test hooks: not part of the drop-in boundary

OK, ERR_ARG, ERR_HIP, ERR_INVARIANT, ERR_NOMEM, ERR_FORMAT = 0, 1, 2, 3, 4, 5


class PcsConfigC(C.Structure):
    _fields_ = [
        ("pow_bits", C.c_uint32),
        ("log_blowup_factor", C.c_uint32),
        ("log_last_layer_degree_bound", C.c_uint32),
        ("n_queries", C.c_uint32),
    ]


def build(force=False):
    """Compile libfrieda_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", os.path.join(_PKG, "csrc"), "-s", "clean"])
    subprocess.check_call(["make", "-C", os.path.join(_PKG, "csrc"), "-s", "-j8"])
    return LIB_PATH


def declared_symbols():
    """Every function name include/frieda_hip.h (the boundary) and include/frieda_hip_testing.h (test hooks) declare."""
    text = open(HEADER_PATH).read() + open(TESTING_HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(frieda_[a-z0-9_]+)\s*\(", text)))


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime.  When this library (linked against /opt/rocm's) is loaded BEFORE torch, the runtime that
    # initialises second fails (torch.cuda.is_available() turns False, or frieda_ctx_create returns FRIEDA_ERR_HIP); with torch loaded
    # first both work.  So a process that has torch gets it imported here, before the library; one without torch is not affected.
    import importlib.util
    import sys

    # (best effort: a broken or half-imported torch must not keep a CPU-only user — declared_symbols, the host verifier — from the library)
    try:
        if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(frieda_amd has no CPU fallback)"
        )
    L = C.CDLL(LIB_PATH)
    vp, sz, u32, u64 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64
    pp = C.POINTER(C.c_void_p)
    u64p = C.POINTER(C.c_uint64)
    sig = {
        "frieda_abi_version": (u32, []),
        "frieda_status_string": (C.c_char_p, [C.c_int]),
        "frieda_last_error": (C.c_char_p, [vp]),
        "frieda_ctx_notes": (C.c_char_p, [vp]),
        "frieda_ctx_create": (C.c_int, [C.c_int, vp, pp]),
        "frieda_ctx_destroy": (C.c_int, [vp]),
        "frieda_ctx_synchronize": (C.c_int, [vp]),
        "frieda_ctx_release_workspace": (C.c_int, [vp]),
        "frieda_ctx_set_twiddle_cache": (C.c_int, [vp, C.c_int]),
        "frieda_ctx_set_host_channel": (C.c_int, [vp, C.c_int]),
        "frieda_ctx_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
        "frieda_ctx_test_set_draw_bound": (C.c_int, [vp, u32]),
        "frieda_ctx_test_set_grind_first_log": (C.c_int, [vp, u32]),
        "frieda_ctx_test_set_arena_limit": (C.c_int, [vp, u64]),
        "frieda_workspace_bytes": (sz, [sz, u32, u32, C.c_int]),
        "frieda_batch_plan": (C.c_int, [vp, sz, u32, u32, C.c_int, u32, u32, C.POINTER(u32), sz, C.POINTER(u32)]),
        "frieda_ctx_set_kernel_timing": (C.c_int, [vp, C.c_int]),
        "frieda_ctx_last_prove_phases": (C.c_int, [vp, C.POINTER(C.c_double)]),
        "frieda_ctx_blake2s_ceiling": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "frieda_ctx_blake2s_ceiling_ex": (C.c_int, [vp, C.POINTER(C.c_double)]),
        "frieda_ctx_last_transcript": (C.c_int, [vp, C.POINTER(u32), vp, sz, vp]),
        "frieda_ctx_kernel_timing_report": (sz, [vp, vp, sz, C.c_int]),
        "frieda_commit": (C.c_int, [vp, vp, sz, u32, vp]),
        "frieda_commit_device": (C.c_int, [vp, vp, sz, u32, vp]),
        "frieda_commit_and_generate_proof": (C.c_int, [vp, vp, sz, u64p, PcsConfigC, vp, pp]),
        "frieda_commit_and_generate_proof_device": (C.c_int, [vp, vp, sz, u64p, PcsConfigC, vp, pp]),
        "frieda_generate_proof": (C.c_int, [vp, vp, sz, u64p, PcsConfigC, pp]),
        "frieda_prove_begin": (C.c_int, [vp, vp, sz, u64p, PcsConfigC]),
        "frieda_prove_begin_device": (C.c_int, [vp, vp, sz, u64p, PcsConfigC]),
        "frieda_prove_finish": (C.c_int, [vp, vp, pp]),
        "frieda_commit_and_generate_proof_batch": (C.c_int, [vp, vp, sz, sz, u32, u64p, PcsConfigC, vp, pp]),
        "frieda_commit_and_generate_proof_batch_device": (C.c_int, [vp, vp, sz, sz, u32, u64p, PcsConfigC, vp, pp]),
        "frieda_prove_batch_begin": (C.c_int, [vp, vp, sz, sz, u32, u64p, PcsConfigC]),
        "frieda_prove_batch_begin_device": (C.c_int, [vp, vp, sz, sz, u32, u64p, PcsConfigC]),
        "frieda_prove_batch_finish": (C.c_int, [vp, u32, vp, pp]),
        "frieda_commit_batch": (C.c_int, [vp, vp, sz, sz, u32, u32, vp]),
        "frieda_commit_batch_device": (C.c_int, [vp, vp, sz, sz, u32, u32, vp]),
        "frieda_multi_create": (C.c_int, [C.POINTER(C.c_int), u32, pp]),
        "frieda_multi_destroy": (C.c_int, [vp]),
        "frieda_multi_device_count": (u32, [vp]),
        "frieda_multi_last_error": (C.c_char_p, [vp]),
        "frieda_multi_uses_rccl": (C.c_int, [vp]),
        "frieda_multi_gather_count": (u64, [vp]),
        "frieda_multi_ctx": (vp, [vp, u32]),
        "frieda_multi_release_workspace": (C.c_int, [vp]),
        "frieda_multi_near_cpus": (u32, [vp, u32, C.POINTER(C.c_int), sz]),
        "frieda_test_near_cpus": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), sz, C.POINTER(sz)]),
        "frieda_test_parse_cpulist": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), sz, C.POINTER(sz)]),
        "frieda_commit_many": (C.c_int, [vp, pp, C.POINTER(sz), u32, u32, vp]),
        "frieda_prove_many": (C.c_int, [vp, pp, C.POINTER(sz), u32, u64p, PcsConfigC, vp, pp]),
        "frieda_verify": (C.c_int, [vp, u64p, C.POINTER(C.c_int)]),
        "frieda_verify_samples": (C.c_int, [vp, u64p, C.POINTER(C.c_int), vp, sz, C.POINTER(sz)]),
        "frieda_proof_free": (None, [vp]),
        "frieda_proof_clone": (C.c_int, [vp, pp]),
        "frieda_proof_proof_of_work": (u64, [vp]),
        "frieda_proof_set_proof_of_work": (None, [vp, u64]),
        "frieda_proof_pcs_config": (PcsConfigC, [vp]),
        "frieda_proof_log_size_bound": (u32, [vp]),
        "frieda_proof_n_evaluations": (sz, [vp]),
        "frieda_proof_evaluations": (C.POINTER(u32), [vp]),
        "frieda_proof_resize_evaluations": (C.c_int, [vp, sz]),
        "frieda_proof_n_inner_layers": (sz, [vp]),
        "frieda_proof_layer_commitment": (C.POINTER(C.c_uint8), [vp, sz]),
        "frieda_proof_layer_fri_witness": (C.POINTER(u32), [vp, sz, C.POINTER(sz)]),
        "frieda_proof_layer_hash_witness": (C.POINTER(C.c_uint8), [vp, sz, C.POINTER(sz)]),
        "frieda_proof_layer_column_witness": (C.POINTER(u32), [vp, sz, C.POINTER(sz)]),
        "frieda_proof_last_layer_poly": (C.POINTER(u32), [vp, C.POINTER(sz)]),
        "frieda_proof_serialize": (sz, [vp, vp, sz]),
        "frieda_proof_deserialize": (C.c_int, [vp, sz, pp]),
        "frieda_dev_alloc": (C.c_int, [vp, sz, pp]),
        "frieda_dev_free": (C.c_int, [vp, vp]),
        "frieda_dev_upload": (C.c_int, [vp, vp, vp, sz]),
        "frieda_dev_download": (C.c_int, [vp, vp, vp, sz]),
        "frieda_dev_at": (C.c_int, [vp, vp, sz, C.POINTER(u32)]),
        "frieda_dev_at_secure": (C.c_int, [vp, vp, sz, sz, C.POINTER(u32)]),
        "frieda_bit_reverse_column": (C.c_int, [vp, vp, sz, u32, u32]),
        "frieda_codec_shape": (C.c_int, [sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(u32)]),
        "frieda_unpack30": (C.c_int, [vp, vp, sz, vp, sz]),
        "frieda_precompute_twiddles": (C.c_int, [vp, u32, pp, pp]),
        "frieda_circle_evaluate": (C.c_int, [vp, vp, u32, u32, u32, vp]),
        "frieda_circle_interpolate": (C.c_int, [vp, vp, u32, u32, u32, u32, vp]),
        "frieda_pack30": (C.c_int, [vp, vp, sz, vp, sz]),
        "frieda_reconstruct_device": (C.c_int, [vp, vp, u32, u32, u32, sz, vp]),
        "frieda_circle_interpolate_cells": (C.c_int, [vp, vp, vp, u32, u32, u32, u32, u32, vp]),
        "frieda_circle_interpolate_cells_any": (C.c_int, [vp, vp, vp, u32, u32, u32, u32, u32, vp, vp]),
        "frieda_reconstruct_cells_device": (C.c_int, [vp, vp, vp, u32, u32, u32, u32, sz, vp]),
        "frieda_circle_interpolate_points": (C.c_int, [vp, vp, vp, u32, u32, u32, u32, u32, vp]),
        "frieda_reconstruct_points_device": (C.c_int, [vp, vp, vp, u32, u32, u32, u32, sz, vp]),
        "frieda_merkle_commit_layer": (C.c_int, [vp, u32, vp, pp, u32, vp]),
        "frieda_merkle_commit": (C.c_int, [vp, vp, u32, vp]),
        "frieda_merkle_layer_offset": (sz, [u32, u32]),
        "frieda_merkle_root": (C.c_int, [vp, vp, u32, vp]),
        "frieda_fold_circle_into_line": (C.c_int, [vp, vp, vp, u32, vp]),
        "frieda_fold_line": (C.c_int, [vp, vp, u32, u32, vp, vp]),
        "frieda_circle_evaluate_fold2": (C.c_int, [vp, vp, u32, u32, vp, vp, C.c_int, vp, vp, vp]),
        "frieda_circle_extend": (C.c_int, [vp, vp, u32, u32, u32, vp]),
        "frieda_circle_eval_at_point": (C.c_int, [vp, vp, u32, u32, vp, vp, vp]),
        "frieda_fri_decompose": (C.c_int, [vp, vp, u32, vp, vp]),
        "frieda_grind": (C.c_int, [vp, vp, u32, u64p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    L._signatures = sig
    _lib = L
    return L
