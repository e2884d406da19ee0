"""Does giving every in-flight batch its own part of the chip (HIP CU-masked streams) beat sharing all of it?  P contexts on streams masked to
256 / P CUs each (contiguous or interleaved CU numbers), `batch` blobs per call, one call in flight per context; ms per blob over K distinct
2^n-domain blobs, every proof verified.  P = 0: the product's own BatchPipeline (unmasked streams, 2 in flight).  Measurement aid.
usage: python tools/cu_partition_probe.py [log_domain] [blobs] [batch]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, frieda_amd
from bench import blob_len_for, splitmix64_bytes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
BSZ = int(sys.argv[3]) if len(sys.argv) > 3 else 4
hip = C.CDLL("libamdhip64.so")
blob_len = blob_len_for(n)
blobs = torch.empty((K, blob_len), dtype=torch.uint8, device="cuda")
for i in range(K):
    blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + i, blob_len)))
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
n_cu = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(cus):
    words = (C.c_uint32 * ((n_cu + 31) // 32))()
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), len(words), words)
    assert rc == 0, rc
    return s


def run(ctxs):
    inflight, free, out = [], list(ctxs), []
    for i in range(0, K, BSZ):
        cnt = min(BSZ, K - i)
        if not free:
            c, k = inflight.pop(0)
            out.extend(c.prove_batch_finish(k))
            free.append(c)
        c = free.pop(0)
        c.prove_batch_begin_device(blobs[i].data_ptr(), blob_len, blob_len, cnt, [blob_len] * cnt, cfg)
        inflight.append((c, cnt))
    for c, k in inflight:
        out.extend(c.prove_batch_finish(k))
    return out


ref = None
CASES = [(0, "-"), (2, "contiguous"), (2, "interleaved"), (4, "contiguous"), (4, "interleaved"), (8, "contiguous"), (8, "interleaved")]
if len(sys.argv) > 4:  # e.g. "0:-,4:interleaved,3:interleaved"
    CASES = [(int(c.split(":")[0]), c.split(":")[1]) for c in sys.argv[4].split(",")]
for P, layout in CASES:
    if P == 0:
        ctxs = [frieda_amd.Context(0), frieda_amd.Context(0)]
    else:
        parts = [[c for c in range(n_cu) if (c * P // n_cu if layout == "contiguous" else c % P) == p] for p in range(P)]
        ctxs = [frieda_amd.Context(0, masked_stream(part).value) for part in parts]
    for _ in range(2):
        run(ctxs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run(ctxs)
    dt = (time.perf_counter() - t0) / K
    roots = [r for r, _ in res]
    ref = ref or roots
    assert roots == ref and all(frieda_amd.verify(p, blob_len) for _, p in res)
    print(f"n={n} batch {BSZ}: {('unmasked, 2 in flight' if P == 0 else f'{P} partitions of {n_cu // P} CUs ({layout})').ljust(40)} {1e3 * dt:.4f} ms per blob", flush=True)
    for c in ctxs:
        c.close()
