import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, frieda_amd
from conftest import splitmix64_bytes
from util import blob_len_for
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
n, count = int(sys.argv[1]), int(sys.argv[2])
blob_len = blob_len_for(n)
blobs = torch.empty((count, blob_len), dtype=torch.uint8, device="cuda")
for i in range(count):
    blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + i, blob_len)))
torch.cuda.synchronize()
ctx = frieda_amd.Context(0)
call = lambda: ctx.commit_and_generate_proof_batch_device(blobs[0].data_ptr(), blob_len, blob_len, count, [blob_len] * count, cfg)
call(); call(); torch.cuda.synchronize(); time.sleep(0.02); call()
