"""Time of the reconstruction entry points on a 2^L-coefficient polynomial (4 columns), by number of cells.  Measurement aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import frieda_amd
from conftest import splitmix64_bytes
from util import DevBuf, blob_len_for

ctx = frieda_amd.Context(0)
L_ = ctx._L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
L = n - 4
data = splitmix64_bytes(5, blob_len_for(n))
d_in = DevBuf.from_array(ctx, data)
d_coef, d_ev = DevBuf(ctx, 16 << L), DevBuf(ctx, 16 << n)
assert L_.frieda_unpack30(ctx._h, d_in.ptr, data.size, d_coef.ptr, 4 << L) == 0
assert L_.frieda_circle_evaluate(ctx._h, d_coef.ptr, 4, L, n, d_ev.ptr) == 0
ev = d_ev.to_array(np.uint32, (4, 1 << n))
rng = np.random.default_rng(1)
print("| cells | cell words | ms |")
print("|---|---|---|")
for j in (0, 2, 4, 6, 8):
    m = L - j
    idx = rng.choice(1 << (n - m), size=1 << j, replace=False).astype(np.uint32)
    cells = np.ascontiguousarray(np.stack([ev[:, int(c) << m : (int(c) + 1) << m] for c in idx]))
    d_cells, d_o = DevBuf.from_array(ctx, cells), DevBuf(ctx, data.size + 8)
    for rep in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        rc = L_.frieda_reconstruct_cells_device(ctx._h, d_cells.ptr, idx.ctypes.data, 1 << j, m, L, n, data.size, d_o.ptr)
        ctx.synchronize()
        dt = time.perf_counter() - t0
    assert rc == 0 and d_o.to_array(np.uint8, (data.size,)).tobytes() == data.tobytes()
    print(f"| {1 << j} | 2^{m} | {1e3 * dt:.2f} |")

# any >= 2^L + 2 points (frieda_reconstruct_points_device: erasure-locator route, no bound on the number of cells)
print("\n| points path: cells | cell words | spare cells | ms |")
print("|---|---|---|---|")
for m, extra in ((0, 2), (0, 1 << (L - 2)), (4, 8), (8, 2), (L - 4, 1)):
    if m > L:
        continue
    n_cells = (1 << (L - m)) + extra
    idx = rng.permutation(1 << (n - m))[:n_cells].astype(np.uint32)
    cells = np.ascontiguousarray(ev.reshape(4, -1, 1 << m)[:, idx, :].transpose(1, 0, 2))
    d_cells, d_o = DevBuf.from_array(ctx, cells), DevBuf(ctx, data.size + 8)
    for rep in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        rc = L_.frieda_reconstruct_points_device(ctx._h, d_cells.ptr, idx.ctypes.data, n_cells, m, L, n, data.size, d_o.ptr)
        ctx.synchronize()
        dt = time.perf_counter() - t0
    assert rc == 0 and d_o.to_array(np.uint8, (data.size,)).tobytes() == data.tobytes()
    print(f"| {n_cells} | 2^{m} | {extra} | {1e3 * dt:.2f} |")
    if os.environ.get("KERNEL_REPORT") and m == 0 and extra == 2:  # where the time of the single-point case goes (device side)
        ctx.set_kernel_timing(True)
        L_.frieda_reconstruct_points_device(ctx._h, d_cells.ptr, idx.ctypes.data, n_cells, m, L, n, data.size, d_o.ptr)
        rep = ctx.kernel_timing_report()
        ctx.set_kernel_timing(False)
        ks = rep if isinstance(rep, list) else rep["kernels"]
        print("  device spans (ms): " + ", ".join(f"{k['name']} x{k['launches']} {k['total_ms']:.3f}" for k in ks) + f"; sum {sum(k['total_ms'] for k in ks):.3f}")
