# lone proofs and commits, stream of proofs / commits, with the fused last pass + tree launch against the two separate launches
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, frieda_amd
from conftest import splitmix64_bytes
from util import blob_len_for
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
vals = [0, 1]
for n in (24, 22, 21, 20, 19, 18):
    blob_len = blob_len_for(n)
    blob = torch.from_numpy(splitmix64_bytes(100, blob_len)).cuda(); torch.cuda.synchronize()
    for what in ("prove", "commit"):
        row = []
        for v in vals:
            ctx = frieda_amd.Context(0); ctx.set_option("FRIEDA_NO_ENCODE_TREE_FUSION", v)
            if what == "prove":
                f = lambda: ctx.commit_and_generate_proof_device(blob.data_ptr(), blob_len, blob_len, cfg)
            else:
                root = torch.empty(32, dtype=torch.uint8, device="cuda")
                f = lambda: ctx.commit_device(blob.data_ptr(), blob_len, 4, root.data_ptr())
            for _ in range(3): f()
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10): f()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 10)
            row.append(best); ctx.close()
        print(f"lone {what} 2^{n}: fused {1e3*row[0]:.4f}  separate {1e3*row[1]:.4f} ms", flush=True)
