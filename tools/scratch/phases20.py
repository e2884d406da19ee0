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
for cut in ([10, 10], [5, 5, 5, 5], [20]):
    pipe = frieda_amd.BatchPipeline(0, 2)
    def run(log=False):
        T = [time.perf_counter()]; i = 0; out = []
        for cnt in cut:
            r = pipe.submit_device(blobs[i].data_ptr(), blob_len, blob_len, cnt, [blob_len]*cnt, cfg)
            T.append(time.perf_counter())
            if r is not None: out.extend(r)
            i += cnt
        while pipe.inflight:
            ctx, cnt = pipe.inflight.pop(0)
            out.extend(ctx.prove_batch_finish(cnt)); pipe.free.append(ctx)
            T.append(time.perf_counter())
            if log: print("   finish phases(ms):", {k: round(v, 3) for k, v in ctx.last_prove_phases().items()})
        if log: print("  host marks (ms since start):", [round(1e3*(t-T[0]),3) for t in T[1:]])
        return out
    run(); run()
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(); dt = time.perf_counter()-t0
    print(cut, f"{1e3*dt:.3f} ms total, {1e3*dt/K:.4f} per blob")
    run(True)
    pipe.close()
