"""Probe: how much do the NTT and the first-tree kernels overlap when issued on two streams?  (Decides whether a fused /
chunk-overlapped encode+hash is worth building.)  Measurement aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, frieda_amd
from frieda_amd.api import _check
n, L = 24, 20
a, b = frieda_amd.Context(0), frieda_amd.Context(0)
coef = torch.randint(0, 2**31 - 1, (4, 1 << L), dtype=torch.int32, device="cuda")
ev_a = torch.empty((4, 1 << n), dtype=torch.int32, device="cuda")
ev_b = torch.randint(0, 2**31 - 1, (4, 1 << n), dtype=torch.int32, device="cuda")
root = torch.zeros(32, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
def ntt(): _check(a._L.frieda_circle_evaluate(a._h, coef.data_ptr(), 4, L, n, ev_a.data_ptr()), a._h)
def tree(): _check(b._L.frieda_merkle_root(b._h, ev_b.data_ptr(), n, root.data_ptr()), b._h)
for _ in range(3): ntt(); tree()
a.synchronize(); b.synchronize()
def timeit(f, reps=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / reps
t_ntt = timeit(lambda: (ntt(), a.synchronize()))
t_tree = timeit(lambda: (tree(), b.synchronize()))
t_both = timeit(lambda: (ntt(), tree(), a.synchronize(), b.synchronize()))
print(f"ntt {t_ntt:.3f} ms  tree {t_tree:.3f} ms  sum {t_ntt + t_tree:.3f}  concurrent {t_both:.3f} ms")
