"""Randomised run of the reconstruction side: encode random blobs on the device, sample them the way a client would (a block, scattered
cells, single points; sometimes one corrupted word), and rebuild — the bytes must be the original, a corrupted sample must be reported.
Both locator routes of the points entry are used (FRIEDA_ERASURE_TREE_MIN_LOG is flipped per case).  Not part of the test suite.
usage: python tools/fuzz_reconstruct.py [seconds] [seed]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import frieda_amd
from conftest import splitmix64_bytes
from util import DevBuf


def encode(ctx, data, B):
    L_ = ctx._L
    nf, npad, lg = C.c_size_t(), C.c_size_t(), C.c_uint32()
    L_.frieda_codec_shape(len(data), C.byref(nf), C.byref(npad), C.byref(lg))
    L, n = lg.value, lg.value + B
    d_in = DevBuf.from_array(ctx, np.frombuffer(data, dtype=np.uint8))
    d_coef, d_ev = DevBuf(ctx, 4 * npad.value), DevBuf(ctx, 16 << n)
    assert L_.frieda_unpack30(ctx._h, d_in.ptr, len(data), d_coef.ptr, npad.value) == 0
    assert L_.frieda_circle_evaluate(ctx._h, d_coef.ptr, 4, L, n, d_ev.ptr) == 0
    return d_ev.to_array(np.uint32, (4, 1 << n)), L, n


def run(budget, seed, ctx=None):
    """-> dict of case counts"""
    rng = random.Random(seed)
    nrng = np.random.default_rng(seed)
    ctx = ctx or frieda_amd.Context(0)
    counts = {"block": 0, "cells_dense": 0, "points_lines": 0, "points_tree": 0, "points_cells": 0, "corrupt_reported": 0}
    t0 = t_print = time.time()
    while time.time() - t0 < budget:
        if time.time() - t_print > 30:
            t_print = time.time()
            print(f"  ... {counts} after {t_print - t0:.0f} s", flush=True)
        B = rng.choice([1, 2, 3, 4, 4, 5])
        size = rng.choice([rng.randint(1, 300), rng.randint(300, 20000), rng.randint(20000, 400000), rng.randint(400000, 4000000)])
        data = splitmix64_bytes(rng.randint(1, 1 << 30), size).tobytes()
        ev, L, n = encode(ctx, data, B)
        if n > 23:
            continue
        # one aligned block
        k = rng.randrange(1 << B)
        blk = np.ascontiguousarray(ev[:, k << L : (k + 1) << L])
        assert ctx.reconstruct_from_block(blk, n, k, size) == data, ("block", size, B, k)
        counts["block"] += 1
        # scattered cells, dense route (exactly 2^L values, at most 256 cells here)
        if L >= 1:
            m = rng.randint(max(0, L - 8), L)
            idx = nrng.permutation(1 << (n - m))[: 1 << (L - m)].astype(np.uint32)
            cells = np.ascontiguousarray(np.stack([ev[:, int(c) << m : (int(c) + 1) << m] for c in idx]))
            try:
                assert ctx.reconstruct_from_cells(cells, idx, L, n, size) == data, ("cells", size, B, m)
                counts["cells_dense"] += 1
            except frieda_amd.FriedaError as e:  # single points can be a singular system: reported, never a wrong answer
                assert m == 0 and "singular" in str(e), (size, B, m, str(e))
        # any >= 2^L + 2 points
        if L >= 1 and n >= 2:
            m = rng.choice([0, 0, 0, 1, 2, 4, 6])
            m = min(m, n - 1)
            need = ((1 << max(L - m, 0)) + 1) if m > 0 else (1 << L) + 2
            total = 1 << (n - m)
            if need > total:
                continue
            n_cells = min(total, need + rng.choice([0, 0, 1, 7, need // 3]))
            idx = nrng.permutation(total)[:n_cells].astype(np.uint32)
            cells = np.ascontiguousarray(ev.reshape(4, -1, 1 << m)[:, idx, :].transpose(1, 0, 2))
            tree = rng.random() < 0.5
            ctx.set_option("FRIEDA_ERASURE_TREE_MIN_LOG", 6 if tree else 32)
            corrupt = rng.random() < 0.2
            if corrupt:
                cells[rng.randrange(n_cells), rng.randrange(4), rng.randrange(1 << m)] ^= 1 << rng.randrange(30)
                try:
                    ctx.reconstruct_from_points(cells, idx, L, n, size)
                except frieda_amd.FriedaError as e:
                    assert "not values of one polynomial" in str(e), str(e)
                    counts["corrupt_reported"] += 1
                else:
                    raise AssertionError(("a corrupted sample went unnoticed", size, B, m, n_cells, tree))
            else:
                assert ctx.reconstruct_from_points(cells, idx, L, n, size) == data, ("points", size, B, m, n_cells, tree)
                counts["points_cells" if m > 0 else ("points_tree" if tree and L >= 6 else "points_lines")] += 1
    ctx.set_option("FRIEDA_ERASURE_TREE_MIN_LOG", 15)
    return counts, time.time() - t0


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    counts, dt = run(budget, seed)
    print(f"fuzz_reconstruct: {counts} in {dt:.0f} s, seed {seed}: every blob came back byte for byte, every corrupted sample was reported")
