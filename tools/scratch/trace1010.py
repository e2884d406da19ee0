import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, frieda_amd
from conftest import splitmix64_bytes
from util import blob_len_for
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
n = 24; blob_len = blob_len_for(n); K = 20
cut = [int(x) for x in sys.argv[1].split(",")]
blobs = torch.empty((K, blob_len), dtype=torch.uint8, device="cuda")
for i in range(K):
    blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + i, blob_len)))
torch.cuda.synchronize()
pipe = frieda_amd.BatchPipeline(0, 2)
def run():
    i = 0
    for cnt in cut:
        pipe.submit_device(blobs[i].data_ptr(), blob_len, blob_len, cnt, [blob_len]*cnt, cfg); i += cnt
    pipe.drain()
run(); run()
torch.cuda.synchronize(); time.sleep(0.05)
t0 = time.perf_counter(); run(); print("timed", 1e3*(time.perf_counter()-t0)/K)
