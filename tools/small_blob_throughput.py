"""Throughput of many small independent blobs (the reference's bench sizes) with several proofs in flight
(frieda_amd.ProofPipeline, one context per in-flight proof).  Measurement aid; prints a markdown table."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import frieda_amd
from conftest import pattern_bytes

cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
print("| blob bytes | depth | proofs/s | us per proof |")
print("|---|---|---|---|")
for size in (1024, 16384, 65536):
    blob = torch.from_numpy(pattern_bytes(size)).cuda()
    for depth in (1, 2, 4, 8):
        pipe = frieda_amd.ProofPipeline(0, depth)
        for _ in range(2 * depth + 2):
            pipe.submit_device(blob.data_ptr(), size, size, cfg)
        pipe.drain()
        torch.cuda.synchronize()
        n = 400
        t0 = time.perf_counter()
        done = 0
        for _ in range(n):
            if pipe.submit_device(blob.data_ptr(), size, size, cfg) is not None:
                done += 1
        done += len(pipe.drain())
        dt = time.perf_counter() - t0
        assert done == n
        print(f"| {size} | {depth} | {n / dt:.0f} | {1e6 * dt / n:.1f} |")
        pipe.close()
