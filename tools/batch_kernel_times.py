"""Per-kernel HIP-event times of one batched prove (frieda_ctx_set_kernel_timing).  usage: python tools/batch_kernel_times.py SIZE COUNT"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import frieda_amd
from conftest import splitmix64_bytes

size, count = int(sys.argv[1]), int(sys.argv[2])
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
ctx = frieda_amd.Context(0)
host = np.concatenate([splitmix64_bytes(31 * i + size, size) for i in range(count)])
dev = torch.from_numpy(host).cuda()
seeds = list(range(count))
for _ in range(2):
    ctx.commit_and_generate_proof_batch_device(dev.data_ptr(), size, size, count, seeds, cfg)
ctx.set_kernel_timing(True)
reps = 3
for _ in range(reps):
    ctx.commit_and_generate_proof_batch_device(dev.data_ptr(), size, size, count, seeds, cfg)
kernels = ctx.kernel_timing_report()
ctx.set_kernel_timing(False)
tot = 0.0
for k in sorted(kernels, key=lambda k: -k["total_ms"]):
    print(f"{k['name']:20s} launches/batch {k['launches'] / reps:6.1f}  ms/batch {k['total_ms'] / reps:8.3f}  us/proof {1e3 * k['total_ms'] / reps / count:7.2f}")
    tot += k["total_ms"] / reps
print(f"total {tot:.3f} ms/batch, {1e3 * tot / count:.2f} us/proof")
