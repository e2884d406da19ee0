"""frieda_unpack30 on one blob of a 2^n domain: microseconds and GB/s (read + written).  FRIEDA_UNPACK_TILES=1|2|4|8 for the A/B."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, frieda_amd
ctx = frieda_amd.Context(0); lib, h = ctx._L, ctx._h
for n in (16, 20, 22, 24):
    L = n - 4
    blob_len = (4 << L) * 30 // 8
    data = torch.randint(0, 256, (blob_len,), dtype=torch.uint8, device="cuda")
    out = torch.empty(4 << L, dtype=torch.int32, device="cuda")
    for _ in range(5): assert lib.frieda_unpack30(h, data.data_ptr(), blob_len, out.data_ptr(), 4 << L) == 0
    ctx.synchronize(); reps = 200; t0 = time.perf_counter()
    for _ in range(reps): lib.frieda_unpack30(h, data.data_ptr(), blob_len, out.data_ptr(), 4 << L)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"n={n}: {1e6*dt:.1f} us  {(blob_len + (16 << L)) / dt / 1e9:.0f} GB/s")
