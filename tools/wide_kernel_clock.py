"""In-kernel clock of the two chip-filling product kernels (diagnostic build only): two seconds of lone 2^n proofs, then the stamps
wave 0 of every workgroup of the last launches left behind (clock_stamps.h).  Run on the GPU box:
    tools/build_variant.sh clock -DFRIEDA_CLOCK_STAMPS && FRIEDA_HIP_LIB=build_exp/clock/libfrieda_hip.so python tools/wide_kernel_clock.py [n]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import frieda_amd
from frieda_amd import _lib
from conftest import splitmix64_bytes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
L = _lib.lib()
assert hasattr(L, "frieda_debug_clock_tree5r"), "needs the -DFRIEDA_CLOCK_STAMPS build (FRIEDA_HIP_LIB=build_exp/clock/libfrieda_hip.so)"
blob_len = (4 << (n - 4)) * 30 // 8
blob = torch.from_numpy(splitmix64_bytes(100, blob_len)).cuda()
ctx = frieda_amd.Context(0)
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
t0 = time.time(); k = 0
while time.time() - t0 < secs:
    ctx.commit_and_generate_proof_device(blob.data_ptr(), blob_len, blob_len, cfg); k += 1
ctx.synchronize()
dt = (time.time() - t0) / k
def read(fn, lo, hi):
    buf = np.zeros(4 * 16384, dtype=np.uint64)
    assert fn(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), buf.size) == 0
    s = buf.reshape(-1, 4)[lo:hi].astype(np.float64)
    dc, dr = s[:, 2] - s[:, 0], s[:, 3] - s[:, 1]
    ok = dr > 0
    ghz = dc[ok] / dr[ok] * 0.1
    return np.median(ghz), np.percentile(ghz, 10), np.percentile(ghz, 90), np.median(dr[ok]) * 0.01
for name, fn, lo, hi in (("ntt_last_tree_kernel (fused encode + leaf launch)", L.frieda_debug_clock_ntt_last_tree, 0, (1 << n) >> 12),
                         ("tree5r_kernel, first FRI layer (workgroups of the 2^%d-leaf launch only)" % (n - 1), L.frieda_debug_clock_tree5r, (1 << (n - 1)) >> 11, (1 << (n - 1)) >> 10)):
    fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_ulong]
    med, p10, p90, wg_us = read(fn, lo, min(hi, 16384))
    print(f"{name}: in-kernel clock {med:.3f} GHz (10th - 90th percentile of the workgroups {p10:.3f} - {p90:.3f}), median workgroup lifetime {wg_us:.1f} us")
print(f"({k} lone 2^{n} proofs in {secs:.1f} s, {1e3 * dt:.3f} ms each; the pure Blake2s chain holds 2.39 GHz: profiles/r03_clock_probe_mi355x.txt)")
