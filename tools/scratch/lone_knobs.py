import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, frieda_amd
from conftest import splitmix64_bytes
from util import blob_len_for
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
sets = [{}, {"FRIEDA_T5_WIDE_LOG": 19}, {"FRIEDA_T5_WIDE_LOG": 20}, {"FRIEDA_T5_WIDE_LOG": 21}, {"FRIEDA_T5_WIDE_LOG": 17}, {"FRIEDA_T5_WIDE_LOG": 16},
        {"FRIEDA_T9_MAX_LOG": 18}, {"FRIEDA_T9_MAX_LOG": 19}, {"FRIEDA_T9_MAX_LOG": 16}, {"FRIEDA_TOP_MAX_LOG": 10}, {"FRIEDA_TOP_MAX_LOG": 11},
        {"FRIEDA_T5_WIDE_LOG": 20, "FRIEDA_T9_MAX_LOG": 19}]
for n in (24, 22, 20):
    blob_len = blob_len_for(n)
    blob = torch.from_numpy(splitmix64_bytes(100, blob_len)).cuda()
    torch.cuda.synchronize()
    ref = None
    for opts in sets:
        ctx = frieda_amd.Context(0)
        try:
            for k, v in opts.items(): ctx.set_option(k, v)
        except Exception as e:
            print(n, opts, "refused"); ctx.close(); continue
        f = lambda: ctx.commit_and_generate_proof_device(blob.data_ptr(), blob_len, blob_len, cfg)
        for _ in range(3): r = f()
        if ref is None: ref = r[0]
        assert r[0] == ref
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): f()
            best = min(best, (time.perf_counter() - t0) / 10)
        print(f"n={n} {opts or 'defaults'}: {1e3*best:.4f} ms per lone proof", flush=True)
        ctx.close()
