import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, frieda_amd
from conftest import splitmix64_bytes
from util import blob_len_for
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
n = 20
blob_len = blob_len_for(n)
for count in (96, 128, 160, 256):
    blobs = torch.empty((2 * count, blob_len), dtype=torch.uint8, device="cuda")
    for i in range(2 * count):
        blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + i, blob_len)))
    torch.cuda.synchronize()
    a, b = frieda_amd.Context(0), frieda_amd.Context(0)
    def both():
        t0 = time.perf_counter()
        a.prove_batch_begin_device(blobs[0].data_ptr(), blob_len, blob_len, count, [blob_len] * count, cfg)
        t1 = time.perf_counter()
        b.prove_batch_begin_device(blobs[count].data_ptr(), blob_len, blob_len, count, [blob_len] * count, cfg)
        t2 = time.perf_counter()
        ra = a.prove_batch_finish(count)
        t3 = time.perf_counter()
        rb = b.prove_batch_finish(count)
        t4 = time.perf_counter()
        return [round(1e3 * (x - t0), 2) for x in (t1, t2, t3, t4)]
    both()
    print(count, "begin0 begin1 finish0 finish1 (ms since start):", both(), both())
    for c in (a, b):
        c.set_kernel_timing(True)
    both()
    for nm, c in (("a", a), ("b", b)):
        k = c.kernel_timing_report(reset=True)
        print("   ", nm, {x["name"]: round(x["total_ms"], 2) for x in sorted(k, key=lambda x: -x["total_ms"])[:6]})
        c.set_kernel_timing(False)
    a.close(); b.close(); del blobs
    torch.cuda.empty_cache()
