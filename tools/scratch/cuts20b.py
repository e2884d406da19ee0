import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, frieda_amd
from conftest import splitmix64_bytes
from util import blob_len_for
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
n = 24; blob_len = blob_len_for(n); K = 20
blobs = torch.empty((K, blob_len), dtype=torch.uint8, device="cuda")
for i in range(K):
    blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + i, blob_len)))
torch.cuda.synchronize()
cuts = [[10,10],[8,12],[20],[9,11],[7,13],[12,8],[6,14]]
def run(pipe, cut):
    out = []; i = 0
    for cnt in cut:
        r = pipe.submit_device(blobs[i].data_ptr(), blob_len, blob_len, cnt, [blob_len]*cnt, cfg)
        if r is not None: out.extend(r)
        i += cnt
    out.extend(pipe.drain()); return out
times = [[] for _ in cuts]
for rnd in range(3):  # a fresh pipeline (fresh arenas) per cut and round: instances of one setting differ by 1 - 3 %
    for ci, cut in enumerate(cuts):
        p = frieda_amd.BatchPipeline(0, 2)
        run(p, cut); run(p, cut)
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); run(p, cut); times[ci].append((time.perf_counter()-t0)/K)
        p.close()
for cut, ts in zip(cuts, times):
    ts.sort(); print(cut, f"best {1e3*ts[0]:.4f} median {1e3*ts[len(ts)//2]:.4f} worst {1e3*ts[-1]:.4f}", flush=True)
