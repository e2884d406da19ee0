"""Soak: the same calls over and over for a few minutes, host RSS and device memory printed as it goes (both must stay flat: proofs are
recycled through the context's pool, workspaces only ever grow to the largest shape seen).  usage: python tools/soak.py [seconds] [log_domain]"""
import os, sys, time, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import frieda_amd
from conftest import splitmix64_bytes

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 22
blob_len = (4 << (n - 4)) * 30 // 8
blobs = [splitmix64_bytes(100 + i, blob_len - 17 * (i % 3)) for i in range(16)]  # three lengths: batched units and single proofs
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 16)
mc = frieda_amd.MultiContext([0])
ctx = frieda_amd.Context(0)
def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6
t0 = time.time(); it = 0; first = None; rec = None
while time.time() - t0 < secs:
    seeds = [it * 16 + i for i in range(16)]
    res = mc.prove_many(blobs, seeds, cfg)
    assert all(frieda_amd.verify(p, s) for (_, p), s in zip(res[:2], seeds[:2]))
    roots = mc.commit_many(blobs, 4)
    assert roots == [r for r, _ in res]
    r1, p1 = ctx.commit_and_generate_proof(blobs[it % 16], it, cfg)
    ok, pos = frieda_amd.verify_samples(p1, it)
    assert ok and r1 == roots[it % 16]
    del res, p1
    if it % 5 == 0:  # the reconstruction side too: a third of the codeword's single points -> the blob (product-tree locator, device-side lists)
        if rec is None:
            L_ = ctx._L
            data = blobs[0]
            d_in = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
            coef = torch.empty(4 << (n - 4), dtype=torch.int32, device="cuda")
            ev = torch.empty((4, 1 << n), dtype=torch.int32, device="cuda")
            assert L_.frieda_unpack30(ctx._h, d_in.data_ptr(), len(data), coef.data_ptr(), 4 << (n - 4)) == 0
            assert L_.frieda_circle_evaluate(ctx._h, coef.data_ptr(), 4, n - 4, n, ev.data_ptr()) == 0
            rec = (ev, torch.empty(len(data) + 8, dtype=torch.uint8, device="cuda"), d_in)
        ev, out_b, d_in = rec
        pos_t = torch.randperm(1 << n)[: (1 << (n - 4)) + 2 + (it % 7) * 1000]
        cells = ev[:, pos_t.cuda()].t().contiguous()
        idx = np.ascontiguousarray(pos_t.numpy().astype(np.uint32))
        assert ctx._L.frieda_reconstruct_points_device(ctx._h, cells.data_ptr(), idx.ctypes.data, idx.size, 0, n - 4, n, d_in.numel(), out_b.data_ptr()) == 0
        ctx.synchronize()
        assert torch.equal(out_b[: d_in.numel()], d_in)
    it += 1
    if it % 20 == 0:
        free, total = torch.cuda.mem_get_info()
        cur = (rss_mb(), (total - free) / 1e6)
        if first is None:
            first = cur
        print(f"iteration {it:5d}  {time.time() - t0:6.1f} s  host RSS {cur[0]:8.1f} MB  device memory in use {cur[1]:9.1f} MB", flush=True)
print(f"soak ok: {it} iterations; host RSS {first[0]:.1f} -> {cur[0]:.1f} MB, device {first[1]:.1f} -> {cur[1]:.1f} MB")
