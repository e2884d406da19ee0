import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, frieda_amd
from conftest import splitmix64_bytes
from util import blob_len_for
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
n = 24; blob_len = blob_len_for(n); K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
blobs = torch.empty((K, blob_len), dtype=torch.uint8, device="cuda")
for i in range(K):
    blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + i, blob_len)))
torch.cuda.synchronize()
cuts = [[5,5,5,5],[10,10],[4,8,8],[3,7,10],[6,7,7],[2,6,6,6],[20],[7,13],[8,12],[1,9,10],[2,9,9],[3,3,7,7]] if K == 20 else [[15]*4,[5,15,20,20],[10,20,15,15],[20,20,20],[30,30],[8,16,18,18]]
pipe = frieda_amd.BatchPipeline(0, 2)
def run(cut):
    out = []; i = 0
    for cnt in cut:
        r = pipe.submit_device(blobs[i].data_ptr(), blob_len, blob_len, cnt, [blob_len]*cnt, cfg)
        if r is not None: out.extend(r)
        i += cnt
    out.extend(pipe.drain()); return out
for cut in cuts:
    assert sum(cut) == K
    run(cut); run(cut)
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(cut); best = min(best, (time.perf_counter()-t0)/K)
    print(K, cut, f"{1e3*best:.4f} ms/blob", flush=True)
