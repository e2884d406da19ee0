"""GPU: the fused small-domain path (tree.hip small_first_kernel: unpack + encode + first tree in one launch for domains of 2^8 .. 2^15
points and polynomials of <= 2^11 coefficients per column — the reference's own 1 KiB .. 16 KiB bench inputs, /root/reference/benches/
commit.rs:6-10, benches/proof.rs:14-21) against the oracle, bit-exact, at every (L, n) it takes and at the shapes just outside it.

Host blobs of that size are not copied to the device (the kernel reads page-locked host memory), device blobs may be unaligned."""
import numpy as np
import pytest

from conftest import splitmix64_bytes
from test_gpu_shapes import exact_len

pytestmark = pytest.mark.gpu


def small_path(L, n):
    """mirror of k::small_domain_shape (ids only)"""
    return 8 <= n <= 15 and L <= 11


CASES = []
for _L in range(0, 13):
    for _n in range(max(_L, 1), 17):
        if _n - _L > 9:
            continue
        # inside the fused path: everything; outside: only its border
        if small_path(_L, _n) or _n in (7, 16) or _L == 12:
            CASES.append(pytest.param(_L, _n - _L, id=f"L{_L}-n{_n}-{'fused' if small_path(_L, _n) else 'general'}"))


def _len_for(L, kind):
    e = exact_len(L)
    if L == 0:
        return {"exact": e, "ragged": 7, "short": 1}[kind]
    lo = exact_len(L - 1) + 1  # shortest length with this L is above half
    return {"exact": e, "ragged": max(lo, e - 1 - (e // 5)), "short": lo}[kind]


@pytest.mark.parametrize("kind", ["exact", "ragged", "short"])
@pytest.mark.parametrize("L,B", CASES)
def test_small_commit_and_proof(gpu_ctx, oracle, L, B, kind):
    """commit() and commit_and_generate_proof() from host memory: root and whole proof equal to the oracle's."""
    import ctypes as C

    import frieda_amd

    length = _len_for(L, kind)
    lgs, nf, npad = C.c_uint32(), C.c_size_t(), C.c_size_t()
    gpu_ctx._L.frieda_codec_shape(length, C.byref(nf), C.byref(npad), C.byref(lgs))
    assert lgs.value == L
    data = splitmix64_bytes(5000 + 16 * L + B, length).tobytes()
    assert gpu_ctx.commit(data, B) == oracle.commit(data, B)
    if L < 1 or L + B < 2:
        return
    for last, nq, pw in ((0, 20, 6), (min(2, L - 1), 7, 0)):
        cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(B, last, nq), pw)
        g_root, g_proof = gpu_ctx.commit_and_generate_proof(data, length, cfg)
        o_root, o_proof = oracle.commit_and_generate_proof(data, length, oracle.make_config(pw, B, last, nq))
        assert g_root == o_root and g_proof.serialize() == o_proof.serialize(), (last, nq)


@pytest.mark.parametrize("L,B", [(7, 4), (9, 4), (9, 0), (8, 5), (3, 8), (9, 3), (11, 4), (10, 0), (11, 1)])
def test_small_device_blobs_unaligned_and_batched(gpu_ctx, oracle, L, B):
    """Device-resident blobs at an odd address (byte loads), and batches of equal-length blobs (one launch for all): equal to the oracle."""
    import torch

    import frieda_amd

    length = exact_len(L) - 5
    cnt = 5
    blobs = [splitmix64_bytes(5100 + 8 * L + i, length) for i in range(cnt)]
    want = [oracle.commit(b.tobytes(), B) for b in blobs]
    pad = 3
    stride = length + 13
    buf = torch.zeros(pad + cnt * stride, dtype=torch.uint8, device="cuda")
    for i, b in enumerate(blobs):
        buf[pad + i * stride : pad + i * stride + length].copy_(torch.from_numpy(b))
    root = torch.zeros(32, dtype=torch.uint8, device="cuda")
    for i in range(cnt):
        gpu_ctx.commit_device(buf.data_ptr() + pad + i * stride, length, B, root.data_ptr())
        gpu_ctx.synchronize()
        assert bytes(root.cpu().numpy()) == want[i]
    assert gpu_ctx.commit_batch_device(buf.data_ptr() + pad, stride, length, cnt, B) == want
    assert gpu_ctx.commit_batch([b.tobytes() for b in blobs], B) == want
    cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(B, 0, 20), 5)
    seeds = [11 * i for i in range(cnt)]
    got = gpu_ctx.commit_and_generate_proof_batch_device(buf.data_ptr() + pad, stride, length, cnt, seeds, cfg)
    for (g_root, g_proof), b, s in zip(got, blobs, seeds):
        o_root, o_proof = oracle.commit_and_generate_proof(b.tobytes(), s, oracle.make_config(5, B, 0, 20))
        assert g_root == o_root and g_proof.serialize() == o_proof.serialize()


@pytest.mark.parametrize("B", [8, 9, 12])
def test_empty_and_one_byte_blobs_on_the_fused_path(gpu_ctx, oracle, B):
    """len = 0 and len = 1 (L = 0: four zero / one non-zero felt) at blow-ups that put the domain inside the fused path: commit() root equal
    to the oracle's; the prover panics as the reference does (L - 1 < last)."""
    import frieda_amd

    for data in (b"", b"\x5a"):
        assert gpu_ctx.commit(data, B) == oracle.commit(data, B)
        with pytest.raises(frieda_amd.FriedaPanic):
            gpu_ctx.commit_and_generate_proof(data, 1, frieda_amd.PcsConfig(frieda_amd.FriConfig(B, 0, 20), 4))
        with pytest.raises(RuntimeError):
            oracle.commit_and_generate_proof(data, 1, oracle.make_config(4, B, 0, 20))


def test_reference_bench_inputs_fused_and_general_path(gpu_ctx):
    """The reference's bench inputs (i % 256 for 1024 / 4096 bytes) through the fused path and, on a context with FRIEDA_NO_SMALL_FUSED
    set, through the general one: the same known roots (tests/golden/vectors.json) and the same proof bytes."""
    import frieda_amd

    general = frieda_amd.Context(0)
    general.set_option("FRIEDA_NO_SMALL_FUSED", 1)
    cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
    known = {1024: "636256479b7a848e6664d0d423ec1f84c3d41788c6f783e99ab4e5d058d8264c", 4096: "1e6e5ced9bc64793a1b87dcca719af859cd34406d02147f6849b8f02c914ec74"}
    try:
        for n in (1024, 4096):
            data = bytes(i % 256 for i in range(n))
            outs = []
            for ctx in (gpu_ctx, general):
                r = ctx.commit(data, 4)
                r2, p = ctx.commit_and_generate_proof(data, n, cfg)
                assert r == r2 and frieda_amd.verify(p, n)
                outs.append((r.hex(), p.serialize()))
            assert outs[0] == outs[1] and outs[0][0] == known[n]
    finally:
        general.close()
