import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, frieda_amd
from conftest import splitmix64_bytes
from util import blob_len_for
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
vals = [0, 256, 512, 768, 1024, 1536, 2048, 4096, 1 << 30]
for n in (24, 22, 20, 18):
    blob_len = blob_len_for(n)
    blob = torch.from_numpy(splitmix64_bytes(100, blob_len)).cuda(); torch.cuda.synchronize()
    row = []
    for v in vals:
        ctx = frieda_amd.Context(0); ctx.set_option("FRIEDA_TP_MIN_WGS", v)
        f = lambda: ctx.commit_and_generate_proof_device(blob.data_ptr(), blob_len, blob_len, cfg)
        for _ in range(3): f()
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): f()
            best = min(best, (time.perf_counter() - t0) / 10)
        row.append(best); ctx.close()
    print(f"lone 2^{n}: " + "  ".join(f"{v if v < (1<<30) else 'inf'}:{1e3*b:.4f}" for v, b in zip(vals, row)), flush=True)
for n, K in ((24, 20), (22, 64), (20, 128)):
    blob_len = blob_len_for(n)
    blobs = torch.empty((K, blob_len), dtype=torch.uint8, device="cuda")
    for i in range(K): blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + i, blob_len)))
    torch.cuda.synchronize()
    row = []
    for v in vals:
        p = frieda_amd.BatchPipeline(0, 2)
        for c in p.ctxs: c.set_option("FRIEDA_TP_MIN_WGS", v)
        run = lambda: p.run_stream_device(blobs[0].data_ptr(), blob_len, blob_len, K, [blob_len] * K, cfg)
        run(); run(); best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); run(); best = min(best, (time.perf_counter() - t0) / K)
        row.append(best); p.close()
    print(f"stream 2^{n} x {K}: " + "  ".join(f"{v if v < (1<<30) else 'inf'}:{1e3*b:.4f}" for v, b in zip(vals, row)), flush=True)
