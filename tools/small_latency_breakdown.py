"""Where a lone small proof spends its time (the reference's bench sizes, benches/proof.rs:14-27): per-kernel HIP-event durations, the host's
phase marks and the wall time per call, host blob in, proof out.   usage: python tools/small_latency_breakdown.py [bytes ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import frieda_amd

sizes = [int(a) for a in sys.argv[1:]] or [1024, 4096]
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
ctx = frieda_amd.Context(0)
for n in sizes:
    data = (np.arange(n, dtype=np.uint64) % 256).astype(np.uint8).tobytes()
    for _ in range(5):
        ctx.commit_and_generate_proof(data, n, cfg)
        ctx.commit(data, 4)
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.commit_and_generate_proof(data, n, cfg)
    t_prove = (time.perf_counter() - t0) / reps * 1e6
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.commit(data, 4)
    t_commit = (time.perf_counter() - t0) / reps * 1e6
    print(f"== {n} bytes: prove {t_prove:.1f} us per call, commit {t_commit:.1f} us per call; host phase marks of the last proof (ms): {ctx.last_prove_phases()}")
    for what in ("prove", "commit"):
        ctx.set_kernel_timing(True)
        for _ in range(20):
            ctx.commit_and_generate_proof(data, n, cfg) if what == "prove" else ctx.commit(data, 4)
        rep = ctx.kernel_timing_report(reset=True)
        ctx.set_kernel_timing(False)
        tot = sum(k["total_ms"] for k in rep) / 20 * 1e3
        print(f"  {what}: {tot:.1f} us of kernels per call:", ", ".join(f"{k['name']} {k['total_ms'] / 20 * 1e3:.1f} ({k['launches'] // 20}x)" for k in rep))
