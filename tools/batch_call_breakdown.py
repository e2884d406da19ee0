#!/usr/bin/env python3
"""batch_call_breakdown.py — where ONE batched call spends its time: per-kernel HIP-event totals of a single
commit_and_generate_proof_batch_device call of `count` blobs on a 2^n domain, one context, nothing else on the chip; and the wall time
of the call.  usage: batch_call_breakdown.py <log_domain> <count> [...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import frieda_amd
from conftest import splitmix64_bytes
from util import blob_len_for

cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
args = [int(a) for a in sys.argv[1:]] or [20, 32]
for n, count in zip(args[0::2], args[1::2]):
    blob_len = blob_len_for(n)
    blobs = torch.empty((count, blob_len), dtype=torch.uint8, device="cuda")
    for i in range(count):
        blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + i, blob_len)))
    torch.cuda.synchronize()
    ctx = frieda_amd.Context(0)
    call = lambda: ctx.commit_and_generate_proof_batch_device(blobs[0].data_ptr(), blob_len, blob_len, count, [blob_len] * count, cfg)
    call()
    res = call()
    nonces = [p.proof_of_work for _, p in res]
    print(f"   nonces: sum {sum(nonces) / 1e6:.2f} M, max {max(nonces) / 1e6:.2f} M (the grind scans every blob up to its nonce: >= sum / 46 G/s = {sum(nonces) / 46e3:.0f} us)")
    del res
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        call()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    ctx.set_kernel_timing(True)
    for _ in range(5):
        call()
    rep = ctx.kernel_timing_report(reset=True)
    ctx.set_kernel_timing(False)
    tot = sum(k["total_ms"] for k in rep) / 5
    print(f"== 2^{n} x {count} blobs in one call: wall {wall:.1f} us = {wall / count:.1f} us per blob; kernels {1e3 * tot:.1f} us")
    for k in sorted(rep, key=lambda k: -k["total_ms"]):
        print(f"   {k['name']:22s} {k['launches'] // 5:3d} launches  {1e3 * k['total_ms'] / 5:9.1f} us")
    ctx.close()
    del blobs
    torch.cuda.empty_cache()
