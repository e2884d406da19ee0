import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, frieda_amd
from conftest import splitmix64_bytes
from util import blob_len_for
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
n = 24; blob_len = blob_len_for(n)
Kmax = 60
blobs = torch.empty((Kmax, blob_len), dtype=torch.uint8, device="cuda")
for i in range(Kmax):
    blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + i, blob_len)))
torch.cuda.synchronize()
for K, D, cut in ((20, 2, [10, 10]), (20, 3, [7, 7, 6]), (20, 3, [4, 8, 8]), (20, 4, [5, 5, 5, 5]), (20, 4, [2, 4, 6, 8]), (20, 3, [3, 7, 10]),
                  (60, 2, [30, 30]), (60, 3, [20, 20, 20]), (60, 3, [10, 20, 30]), (60, 4, [15] * 4), (60, 4, [6, 12, 18, 24]), (60, 3, [10] * 6), (60, 4, [5] * 12)):
    pipe = frieda_amd.BatchPipeline(0, D)
    def run():
        i = 0; out = []
        for cnt in cut:
            r = pipe.submit_device(blobs[i].data_ptr(), blob_len, blob_len, cnt, [blob_len]*cnt, cfg); i += cnt
            if r is not None: out.extend(r)
        out.extend(pipe.drain()); return out
    run(); run()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(); best = min(best, (time.perf_counter()-t0)/K)
    print(f"K={K} in_flight={D} cut={cut}: {1e3*best:.4f} ms/blob", flush=True)
    pipe.close()
