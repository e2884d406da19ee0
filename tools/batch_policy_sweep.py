#!/usr/bin/env python3
"""batch_policy_sweep.py — ms per blob of a stream of K equal-length device-resident blobs through BatchPipeline.run_stream_device
(the library's batch policy) for a grid of the policy's two options, per domain size.  Picks the defaults of
FRIEDA_BATCH_BUDGET_MB / FRIEDA_BATCH_CALLS_PER_CTX (round 5).  usage: batch_policy_sweep.py [log_domain ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import frieda_amd
from conftest import splitmix64_bytes
from util import blob_len_for


def main():
    logs = [int(a) for a in sys.argv[1:]] or [11, 16, 20, 22, 24]
    cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
    head = frieda_amd.workspace_bytes(blob_len_for(24), 4)
    for n in logs:
        blob_len = 1024 if n == 11 else blob_len_for(n)
        K = {24: 40, 22: 64, 20: 128}.get(n, 256)
        blobs = torch.empty((K, blob_len), dtype=torch.uint8, device="cuda")
        for i in range(K):
            blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + i, blob_len)))
        torch.cuda.synchronize()
        ref = None
        for k_used in ([K, K // 2] if n >= 20 else [K]):
            for heads in (2, 4, 5, 8, 16):
                for cpc in (1, 2, 4):
                    pipe = frieda_amd.BatchPipeline(0, 2)
                    pipe.ctxs[0].set_option("FRIEDA_BATCH_BUDGET_MB", (heads * head) >> 20)
                    pipe.ctxs[0].set_option("FRIEDA_BATCH_CALLS_PER_CTX", cpc)
                    cut = pipe.plan(blob_len, k_used, cfg)
                    run = lambda: pipe.run_stream_device(blobs[0].data_ptr(), blob_len, blob_len, k_used, [blob_len] * k_used, cfg)
                    run()
                    run()
                    torch.cuda.synchronize()
                    best = 1e9
                    for _ in range(3):
                        t0 = time.perf_counter()
                        res = run()
                        best = min(best, (time.perf_counter() - t0) / k_used)
                    roots = [r for r, _ in res]
                    if ref is None:
                        ref = roots
                    assert roots == ref[:k_used]
                    print(f"n={n:2d} K={k_used:3d} budget={heads:2d} x 2^24-proof  calls_per_ctx={cpc}  cut={len(cut)} x {max(cut)}/{min(cut)}  {1e3 * best:8.4f} ms/blob", flush=True)
                    pipe.close()
        del blobs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
