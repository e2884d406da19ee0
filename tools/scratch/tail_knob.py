import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import frieda_amd
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
for n in [1024, 4096, 16384, 65536, 262146, 1 << 20, 15 << 20]:
    data = (np.arange(n, dtype=np.uint64) % 256).astype(np.uint8).tobytes()
    row = []
    for knob in (9, 10, 11):
        ctx = frieda_amd.Context(0)
        ctx.set_option("FRIEDA_TAIL_RUN_LOG", knob)
        for _ in range(5):
            ctx.commit_and_generate_proof(data, n, cfg)
        reps = 200 if n < (1 << 20) else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            r = ctx.commit_and_generate_proof(data, n, cfg)
        row.append((time.perf_counter() - t0) / reps * 1e6)
        ctx.close()
    print(n, " ".join(f"tail_run_log={k}: {v:8.1f} us" for k, v in zip((9, 10, 11), row)), flush=True)
