"""commit() on a stream of distinct device-resident 2^n-domain blobs: one context one blob per call (the by_config row), two contexts
alternating, and commit_batch_device with 2 / 4 / 8 / 16 blobs per call on one context.  ms per blob.  Measurement aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, frieda_amd
from bench import blob_len_for, splitmix64_bytes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
blob_len = blob_len_for(n)
blobs = torch.empty((K, blob_len), dtype=torch.uint8, device="cuda")
for i in range(K):
    blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + i, blob_len)))
roots = torch.zeros((K, 32), dtype=torch.uint8, device="cuda")
ctxs = [frieda_amd.Context(0), frieda_amd.Context(0)]

def sync():
    for c in ctxs: c.synchronize()

def run(n_ctx, bsz):
    out = None
    if bsz == 1:
        for i in range(K):
            ctxs[i % n_ctx].commit_device(blobs[i].data_ptr(), blob_len, 4, roots[i].data_ptr())
        sync()
        out = roots.cpu().numpy().tobytes()
    else:  # (the batched entry returns its roots to the host: one synchronisation per call)
        out = b"".join(b"".join(ctxs[0].commit_batch_device(blobs[i].data_ptr(), blob_len, blob_len, min(bsz, K - i), 4)) for i in range(0, K, bsz))
    return out

ref = None
for n_ctx, bsz in ((1, 1), (2, 1), (1, 2), (1, 4), (1, 8), (1, 16)):
    for _ in range(2): run(n_ctx, bsz)
    sync(); t0 = time.perf_counter(); r = run(n_ctx, bsz); dt = (time.perf_counter() - t0) / K
    ref = ref or r
    assert r == ref
    print(f"n={n}: {n_ctx} context(s), {bsz} blob(s) per call: {1e3*dt:.4f} ms per blob")
