"""BASELINE configs[1] (2^20-element NTT + fold_circle_into_line + one fold_line, Level B entry points): per-kernel HIP-event durations and
the back-to-back wall time per pass.   usage: python tools/config2_breakdown.py [log_domain]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import frieda_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = n - 4
ctx = frieda_amd.Context(0)
lib, h = ctx._L, ctx._h
g = torch.Generator(device="cpu").manual_seed(2)
coef = torch.randint(0, 2**31 - 1, (4, 1 << L), dtype=torch.int32, generator=g).cuda()
ev = torch.empty((4, 1 << n), dtype=torch.int32, device="cuda")
l1 = torch.zeros((4, 1 << (n - 1)), dtype=torch.int32, device="cuda")
l2 = torch.empty((4, 1 << (n - 2)), dtype=torch.int32, device="cuda")
a0, a1 = (C.c_uint32 * 4)(11, 22, 33, 44), (C.c_uint32 * 4)(55, 66, 77, 88)
fused = os.environ.get("UNFUSED") is None

def once():
    if fused:
        rc = lib.frieda_circle_evaluate_fold2(h, coef.data_ptr(), L, n, ev.data_ptr(), a0, 0, l1.data_ptr(), a1, l2.data_ptr())
    else:
        rc = lib.frieda_circle_evaluate(h, coef.data_ptr(), 4, L, n, ev.data_ptr())
        rc |= lib.frieda_fold_circle_into_line(h, l1.data_ptr(), ev.data_ptr(), n, a0)
        rc |= lib.frieda_fold_line(h, l1.data_ptr(), n - 1, n, a1, l2.data_ptr())
    assert rc == 0

for _ in range(5):
    once()
ctx.synchronize()
reps = 300
t0 = time.perf_counter()
for _ in range(reps):
    once()
ctx.synchronize()
dt = (time.perf_counter() - t0) / reps
ctx.set_kernel_timing(True)
for _ in range(20):
    once()
rep = ctx.kernel_timing_report(reset=True)
print(f"2^{n}: {dt * 1e6:.1f} us per pass back to back ({'fused folds' if fused else 'separate folds'});", ", ".join(f"{k['name']} {k['total_ms'] / 20 * 1e3:.1f}" for k in rep))
