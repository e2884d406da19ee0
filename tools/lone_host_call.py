"""A lone call from host memory (the reference's signature, data: &[u8]): commit and commit_and_generate_proof of one 2^n-domain blob from
pageable memory, from page-locked memory and from device memory, milliseconds per call.   usage: python tools/lone_host_call.py [log_domain]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import frieda_amd
from conftest import splitmix64_bytes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
blob_len = (4 << (n - 4)) * 30 // 8
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
ctx = frieda_amd.Context(0)
pageable = splitmix64_bytes(100, blob_len)
pinned_t = torch.from_numpy(pageable.copy()).pin_memory()
pinned = pinned_t.numpy()
dev = torch.from_numpy(pageable).cuda()
root_dev = torch.zeros(32, dtype=torch.uint8, device="cuda")

def timed(fn, reps=12):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3

def dev_commit():
    ctx.commit_device(dev.data_ptr(), blob_len, 4, root_dev.data_ptr()); ctx.synchronize()

rows = [("pageable", lambda: ctx.commit(pageable, 4), lambda: ctx.commit_and_generate_proof(pageable, blob_len, cfg)),
        ("page-locked", lambda: ctx.commit(pinned, 4), lambda: ctx.commit_and_generate_proof(pinned, blob_len, cfg)),
        ("device", dev_commit, lambda: ctx.commit_and_generate_proof_device(dev.data_ptr(), blob_len, blob_len, cfg))]
for name, c, p in rows:
    print(f"2^{n} {name:12s} commit {timed(c):.3f} ms   commit_and_generate_proof {timed(p):.3f} ms")
