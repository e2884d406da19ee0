"""Duration of the first (chip-filling) tree launch by size and kind, with the compression count it performs: how far each
launch is from the chip's Blake2s ceiling and what the fixed cost of a launch is.  Uses the per-kernel HIP-event timer on the
proof path at domain sizes 2^20 .. 2^24.  Measurement aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, frieda_amd
from bench import splitmix64_bytes, blob_len_for

cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
print("| domain | kernel | launches/step | us per step |")
print("|---|---|---|---|")
for n in (20, 21, 22, 23, 24):
    blob = torch.from_numpy(splitmix64_bytes(100, blob_len_for(n))).cuda()
    ctx = frieda_amd.Context(0)
    for _ in range(5):
        ctx.commit_and_generate_proof_device(blob.data_ptr(), blob.numel(), blob.numel(), cfg)
    ctx.set_kernel_timing(True)
    reps = 20
    for _ in range(reps):
        ctx.commit_and_generate_proof_device(blob.data_ptr(), blob.numel(), blob.numel(), cfg)
    for k in sorted(ctx.kernel_timing_report(reset=True), key=lambda k: -k["total_ms"]):
        print(f"| 2^{n} | {k['name']} | {k['launches'] / reps:.0f} | {1e3 * k['total_ms'] / reps:.1f} |")
    ctx.close()
