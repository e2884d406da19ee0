import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, frieda_amd
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
for nbytes in (40000, 65536, 131072, 262146, 524288, 1 << 20, 2 << 20, 4 << 20):
    data = (np.arange(nbytes, dtype=np.uint64) % 251).astype(np.uint8).tobytes()
    row = []
    for v in (256, 128, 64, 32, 8, 1):
        ctx = frieda_amd.Context(0); ctx.set_option("FRIEDA_FUSE_MIN_TILES", v)
        r0 = ctx.commit(data, 4)
        for _ in range(5): ctx.commit(data, 4); ctx.commit_and_generate_proof(data, 7, cfg)
        t0 = time.perf_counter()
        for _ in range(100): ctx.commit(data, 4)
        tc = (time.perf_counter() - t0) / 100
        t0 = time.perf_counter()
        for _ in range(50): ctx.commit_and_generate_proof(data, 7, cfg)
        tp = (time.perf_counter() - t0) / 50
        row.append((v, tc, tp, r0)); ctx.close()
    assert len({r[3] for r in row}) == 1
    print(f"{nbytes:8d} B: " + "  ".join(f"min_tiles {v}: commit {1e6*tc:6.1f} prove {1e6*tp:6.1f} us" for v, tc, tp, _ in row), flush=True)
