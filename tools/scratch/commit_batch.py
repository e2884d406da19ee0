import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, frieda_amd
from conftest import splitmix64_bytes
from util import blob_len_for
for n, K, per in ((22, 64, 1), (22, 64, 8), (22, 64, 16), (22, 64, 32), (24, 32, 1), (24, 32, 4), (24, 32, 8), (24, 32, 16), (20, 128, 1), (20, 128, 32), (20, 128, 64)):
    blob_len = blob_len_for(n)
    blobs = torch.empty((K, blob_len), dtype=torch.uint8, device="cuda")
    for i in range(K):
        blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + i, blob_len)))
    roots_dev = torch.zeros(32 * K, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctxs = [frieda_amd.Context(0), frieda_amd.Context(0)]
    def run():
        out = []
        if per == 1:
            for i in range(K):
                ctxs[i & 1].commit_device(blobs[i].data_ptr(), blob_len, 4, roots_dev.data_ptr() + 32 * i)
            for c in ctxs: c.synchronize()
            return bytes(roots_dev.cpu().numpy())
        for j, i in enumerate(range(0, K, per)):
            out += ctxs[0].commit_batch_device(blobs[i].data_ptr(), blob_len, blob_len, min(per, K - i), 4)
        return b"".join(out)
    r0 = run(); run()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = run(); best = min(best, (time.perf_counter() - t0) / K)
    print(f"n={n} K={K} per call {per}: {1e3*best:.4f} ms per commit", flush=True)
    for c in ctxs: c.close()
    del blobs; torch.cuda.empty_cache()
