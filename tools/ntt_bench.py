"""Times frieda_circle_evaluate alone (4 columns, L = n - 4) with HIP-event kernel timing; tuning aid."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, frieda_amd
from frieda_amd.api import _check
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
L = n - 4
ctx = frieda_amd.Context(0)
coef = torch.randint(0, 2**31 - 1, (4, 1 << L), dtype=torch.int32, device="cuda")
out = torch.empty((4, 1 << n), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    _check(ctx._L.frieda_circle_evaluate(ctx._h, coef.data_ptr(), 4, L, n, out.data_ptr()), ctx._h)
ctx.synchronize()
ctx.set_kernel_timing(True)
for _ in range(20):
    _check(ctx._L.frieda_circle_evaluate(ctx._h, coef.data_ptr(), 4, L, n, out.data_ptr()), ctx._h)
for k in ctx.kernel_timing_report():
    print(os.environ.get("FRIEDA_NTT_CPW", "-"), k["name"], "%.1f us" % (1e3 * k["total_ms"] / k["launches"]))
