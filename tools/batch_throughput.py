"""Throughput of batches of small blobs (frieda_commit_and_generate_proof_batch_device / frieda_commit_batch_device) next to the
one-blob-per-call path, device-resident blobs, the reference's bench config (benches/proof.rs:5-12: blowup 16, 20 queries,
20-bit proof of work).  Measurement aid; prints markdown tables.  usage: python tools/batch_throughput.py [sizes...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import frieda_amd
from conftest import splitmix64_bytes

cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
sizes = [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384, 65536, 262144]
ctx = frieda_amd.Context(0)


def timed(fn, reps):
    import gc

    fn()
    torch.cuda.synchronize()
    gc.disable()  # a full collection of the interpreter (40 - 65 ms with torch imported) must not land in a timed loop: profiles/r06_gc_pause.txt
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    gc.enable()
    return dt


ctx2 = frieda_amd.Context(0)


def overlapped(dev, size, count, seeds, rounds):
    """two contexts alternating begin / finish: the planning of one batch runs under the device work of the next"""
    cs = [ctx, ctx2]
    import gc

    # both contexts sized for this call before the clock starts (the second context's first call at a size allocates its workspace and its
    # pinned block: ~0.5 s at 1024 blobs of 64 KiB, which the two-round loop below used to average in: 5 K instead of 29 K proofs per second)
    for c in cs:
        c.prove_batch_begin_device(dev.data_ptr(), size, size, count, seeds, cfg)
        c.prove_batch_finish(count)
    gc.disable()  # (as in timed())
    cs[0].prove_batch_begin_device(dev.data_ptr(), size, size, count, seeds, cfg)
    t0 = time.perf_counter()
    for r in range(rounds):
        cs[(r + 1) & 1].prove_batch_begin_device(dev.data_ptr(), size, size, count, seeds, cfg)
        cs[r & 1].prove_batch_finish(count)
    dt = time.perf_counter() - t0
    cs[rounds & 1].prove_batch_finish(count)
    gc.enable()
    return dt / rounds


print("| blob bytes | batch | prove: proofs/s | us per proof | phases ms (device done / planned / gathered / assembled) | two batches in flight: proofs/s | commit: roots/s | us per root |")
print("|---|---|---|---|---|---|---|---|")
for size in sizes:
    for count in ((1, 8, 64, 256, 1024) if size <= (1 << 20) else (1, 2, 4, 8)):
        if size * count > (64 << 20) and size <= (1 << 20):
            continue
        host = np.concatenate([splitmix64_bytes(31 * i + size, size) for i in range(min(count, 64))])
        host = np.tile(host, (count + 63) // 64)[: size * count]
        dev = torch.from_numpy(host).cuda()
        seeds = list(range(count))
        reps = max(2, min(50, 2048 // count)) if size <= (1 << 20) else 6
        if count == 1:
            tp = timed(lambda: ctx.commit_and_generate_proof_device(dev.data_ptr(), size, 0, cfg), reps)
            tc = timed(lambda: ctx.commit_batch_device(dev.data_ptr(), size, size, 1, 4), reps)
        else:
            tp = timed(lambda: ctx.commit_and_generate_proof_batch_device(dev.data_ptr(), size, size, count, seeds, cfg), reps)
            tc = timed(lambda: ctx.commit_batch_device(dev.data_ptr(), size, size, count, 4), reps)
        ph = ctx.last_prove_phases()
        to = overlapped(dev, size, count, seeds, max(4, reps)) if count > 1 else float("nan")
        print(f"| {size} | {count} | {count / tp:.0f} | {1e6 * tp / count:.1f} | {ph['device_done']:.2f} / {ph['queries']:.2f} / {ph['gathered']:.2f} / {ph['assembled']:.2f} | {count / to:.0f} | {count / tc:.0f} | {1e6 * tc / count:.1f} |")
