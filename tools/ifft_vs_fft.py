"""Forward (frieda_circle_evaluate, L = n) against inverse (frieda_circle_interpolate, block 0) transform times, 1 and 4 columns.  Measurement aid;
FRIEDA_INTT_GENERIC=1 gives the one-column inverse kernel for the A/B."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, frieda_amd
ctx = frieda_amd.Context(0); lib, h = ctx._L, ctx._h
for n in (20, 22, 24):
    for ncols in (1, 4):
        a = torch.randint(0, 2**31 - 1, (ncols, 1 << n), dtype=torch.int32, device="cuda")
        b = torch.empty_like(a)
        def t(fn, reps=20):
            for _ in range(3): assert fn() == 0
            ctx.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): fn()
            ctx.synchronize(); return 1e6 * (time.perf_counter() - t0) / reps
        fwd = t(lambda: lib.frieda_circle_evaluate(h, a.data_ptr(), ncols, n, n, b.data_ptr()))
        inv = t(lambda: lib.frieda_circle_interpolate(h, a.data_ptr(), ncols, n, n, 0, b.data_ptr()))
        gb = 8.0 * ncols * (1 << n) / 1e9
        print(f"n={n} ncols={ncols}: forward {fwd:.1f} us ({gb/fwd*1e6:.0f} GB/s r+w)  inverse {inv:.1f} us ({gb/inv*1e6:.0f} GB/s)")
