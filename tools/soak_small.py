"""Soak of the round-4 small-blob paths: lone host calls of 1 KiB .. 256 KiB (fused small-domain launch from page-locked host memory, the
unfused mid sizes), commit and proof, per-context options flipped now and then; host RSS and device memory must stay flat.
usage: python tools/soak_small.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import frieda_amd
from conftest import splitmix64_bytes

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
sizes = [100, 300, 1024, 4096, 7000, 16384, 30000, 65536, 262146]
blobs = [splitmix64_bytes(7 + i, s).tobytes() for i, s in enumerate(sizes)]
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 10)
ctx = frieda_amd.Context(0)
def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6
t0 = time.time(); it = 0; first = None; t_print = t0
want = {}
while time.time() - t0 < secs:
    ctx.set_option("FRIEDA_NO_SMALL_FUSED", 1 if it % 7 == 3 else 0)
    ctx.set_option("FRIEDA_HOST_DECOMMIT", 1 if it % 11 == 5 else 0)
    for i, b in enumerate(blobs):
        r = ctx.commit(b, 4)
        r2, p = ctx.commit_and_generate_proof(b, i, cfg)
        assert r == r2 and frieda_amd.verify(p, i)
        key = (i, r, p.serialize())
        assert want.setdefault(i, key) == key  # the same bytes whatever the options
        del p
    it += 1
    if first is None and it == 20:
        first = (rss_mb(), torch.cuda.mem_get_info()[0] / 1e6)
    if time.time() - t_print > 20:
        t_print = time.time()
        print(f"  {it} rounds of {len(blobs)} blobs, host RSS {rss_mb():.0f} MB, device free {torch.cuda.mem_get_info()[0] / 1e6:.0f} MB", flush=True)
last = (rss_mb(), torch.cuda.mem_get_info()[0] / 1e6)
print(f"soak_small ok: {it} rounds in {time.time() - t0:.0f} s; host RSS {first[0]:.0f} -> {last[0]:.0f} MB, device free {first[1]:.0f} -> {last[1]:.0f} MB (after round 20 vs at the end)")
assert last[0] - first[0] < 64 and first[1] - last[1] < 64
