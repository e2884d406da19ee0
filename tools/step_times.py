"""Per-step wall time of the first proofs in a fresh process (warm-up behaviour). Measurement aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np, frieda_amd
from bench import splitmix64_bytes, blob_len_for
n = 24
blob = torch.from_numpy(splitmix64_bytes(100, blob_len_for(n))).cuda()
s = torch.cuda.Stream()
ctx = frieda_amd.Context(0, s.cuda_stream)
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
torch.cuda.synchronize()
ts = []
for i in range(40):
    t0 = time.perf_counter()
    ctx.commit_and_generate_proof_device(blob.data_ptr(), blob.numel(), blob.numel(), cfg)
    ts.append(1e3 * (time.perf_counter() - t0))
print(" ".join(f"{t:.2f}" for t in ts))
