import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, frieda_amd
from conftest import splitmix64_bytes
from util import blob_len_for
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
head = frieda_amd.workspace_bytes(blob_len_for(24), 4)
for n, Ks in ((24, (20, 60, 120)), (22, (64, 16)), (20, (64, 128))):
    blob_len = blob_len_for(n)
    Kmax = max(Ks)
    blobs = torch.empty((Kmax, blob_len), dtype=torch.uint8, device="cuda")
    for i in range(Kmax):
        blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + i, blob_len)))
    torch.cuda.synchronize()
    for K in Ks:
        for heads in (5, 8, 10, 16, 20):
            pipe = frieda_amd.BatchPipeline(0, 2)
            pipe.ctxs[0].set_option("FRIEDA_BATCH_BUDGET_MB", ((heads * head) >> 20) + 1)
            cut = pipe.plan(blob_len, K, cfg)
            run = lambda: pipe.run_stream_device(blobs[0].data_ptr(), blob_len, blob_len, K, [blob_len] * K, cfg)
            run(); run()
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter(); run(); best = min(best, (time.perf_counter() - t0) / K)
            print(f"n={n} K={K:3d} budget={heads:2d} heads cut={len(cut)} x {max(cut)}/{min(cut)}  {1e3*best:.4f} ms/blob", flush=True)
            pipe.close()
    del blobs; torch.cuda.empty_cache()
