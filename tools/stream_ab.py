#!/usr/bin/env python3
"""stream_ab.py — A/B of per-context options on the measured loop itself: K device-resident blobs of a 2^n domain through
BatchPipeline.run_stream_device (library batch policy, 2 calls in flight), alternating the option sets over several fresh pipeline instances each; best and median.
(Memory: every instance keeps its workspaces — up to 16 headline proofs per context; add FRIEDA_BATCH_BUDGET_MB to the sets to stay within the HBM.)
usage: stream_ab.py <log_domain>[,<log_domain>...] NAME=VALUE[,NAME=VALUE...] [more option sets ...]   ("-" = defaults)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import frieda_amd
from conftest import splitmix64_bytes
from util import blob_len_for


def main():
    logs = [int(x) for x in sys.argv[1].split(",")]
    sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")) if a != "-" else {} for a in (sys.argv[2:] or ["-"])]
    cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
    for n in logs:
        blob_len = blob_len_for(n)
        K = {24: 40, 23: 40, 22: 64, 21: 64, 20: 128}.get(n, 256)
        blobs = torch.empty((K, blob_len), dtype=torch.uint8, device="cuda")
        for i in range(K):
            blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + i, blob_len)))
        torch.cuda.synchronize()
        INST = 2  # fresh pipelines per option set: instances of the SAME options differ by 1 - 3 % (where their arenas land), so every set
        # gets several and the report is the best and the median over instances x rounds
        pipes = []
        for opts in sets:
            for _ in range(INST):
                p = frieda_amd.BatchPipeline(0, 2)
                for c in p.ctxs:
                    for k, v in opts.items():
                        c.set_option(k, v)
                pipes.append(p)
        times = [[] for _ in sets]
        ref = None
        for rnd in range(4):
            for i, p in enumerate(pipes):
                t0 = time.perf_counter()
                res = p.run_stream_device(blobs[0].data_ptr(), blob_len, blob_len, K, [blob_len] * K, cfg)
                dt = (time.perf_counter() - t0) / K
                roots = [r for r, _ in res]
                if ref is None:
                    ref = roots
                assert roots == ref
                if rnd:
                    times[i // INST].append(dt)
        for opts, ts in zip(sets, times):
            ts.sort()
            print(f"n={n} K={K} {opts or 'defaults'}: best {1e3 * ts[0]:.4f}  median {1e3 * ts[len(ts) // 2]:.4f} ms/blob  ({len(ts)} runs over {INST} instances)", flush=True)
        for p in pipes:
            p.close()
        del blobs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
