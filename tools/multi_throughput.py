"""Throughput of the C ABI's multi-GPU entry (frieda_prove_many / frieda_commit_many) with HOST blobs — the PCIe-inclusive rate a
Rust caller of the drop-in sees.  usage: python tools/multi_throughput.py [log_domain] [n_blobs] [devices, comma separated] [pageable|pinned]
pinned: the blobs sit in page-locked host memory (torch pin_memory = hipHostMalloc), so every upload is an asynchronous DMA."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import frieda_amd
from conftest import splitmix64_bytes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
devices = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
blob_len = (4 << (n - 4)) * 30 // 8
mode = sys.argv[4] if len(sys.argv) > 4 else "pageable"
blobs = [splitmix64_bytes(100 + i, blob_len) for i in range(count)]  # pageable host memory
if mode == "pinned":
    import torch
    _keep = [torch.from_numpy(b).pin_memory() for b in blobs]
    blobs = [t.numpy() for t in _keep]
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
seeds = [blob_len] * count
mc = frieda_amd.MultiContext(devices)
# the first pass sizes the workspaces (the batch policy's units: up to 43 GB per context at 2^24), the upload ring and the twiddles;
# the better of the two passes after it is reported.  FRIEDA_BATCH_BUDGET_MB / FRIEDA_BATCH_CALLS_PER_CTX in the environment change the cut.
mc.prove_many(blobs, seeds, cfg)
mc.commit_many(blobs, 4)
dt = dtc = 1e9
for _ in range(2):
    t0 = time.perf_counter()
    res = mc.prove_many(blobs, seeds, cfg)
    dt = min(dt, time.perf_counter() - t0)
    assert len({r for r, _ in res}) == count and all(frieda_amd.verify(p, s) for (_, p), s in zip(res, seeds))
    t1 = time.perf_counter()
    roots = mc.commit_many(blobs, 4)
    dtc = min(dtc, time.perf_counter() - t1)
    assert roots == [r for r, _ in res]
    del res
el = 4.0 * (1 << n)
print(f"frieda_prove_many : {count} {mode} host blobs of 2^{n} on devices {devices}: {1e3 * dt / count:.3f} ms per blob, {el * count / dt / 1e9:.2f} G M31 elems/s "
      f"(H2D of {blob_len / 1e6:.1f} MB per blob included; rccl={mc.uses_rccl})")
print(f"frieda_commit_many: {1e3 * dtc / count:.3f} ms per blob, {el * count / dtc / 1e9:.2f} G M31 elems/s")
mc.close()
